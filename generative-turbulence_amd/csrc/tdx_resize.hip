// Trilinear resize, align_corners=True, NDHWC (ddpm.py:359-361, 367-369).
// HBM-bound gather kernels: one lane owns 8 channels of one voxel (16 B bf16 / 32 B f32).
// A workgroup owns a compact 4 x 8 x 8 tile of the tensor it WRITES, so that the voxels it gathers
// from form a small 3D region that stays in the CU's L1 (a linear thread->voxel map re-fetched every
// gathered voxel ~17x from L2); all loads of a thread are issued before the arithmetic.
// Index arithmetic follows ATen's upsample_trilinear3d exactly (float scale, float source
// index, truncation, clamped second tap) so that tap selection matches the reference.
#include "tdx_common.h"
#include <stdlib.h>

struct AxisMap {
    int in, out;
    float scale;
};
__host__ __device__ inline AxisMap make_axis(int in, int out) {
    AxisMap a;
    a.in = in; a.out = out;
    a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    return a;
}
// taps of output index o: i0, i1, weight of i1
__host__ __device__ __forceinline__ void axis_taps(const AxisMap& a, int o, int& i0, int& i1, float& w1) {
    const float src = a.scale * (float)o;
    i0 = min((int)src, a.in - 1);
    w1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
    i1 = i0 + ((i0 + 1 < a.in) ? 1 : 0);
}

// tile of the written grid owned by one workgroup: tx x ty x tz voxels (powers of two, <= 4 x 8 x 8),
// shrunk for wide channel counts so that a workgroup makes at most ~8 passes of 256 lanes
struct TileGrid {
    int nx, ny, nz;     // tiles per axis
    int sy, sz;         // log2(ty), log2(tz)
    int tx, ty, tz;
};
#define RS_MAXT 8   // largest tile extent along an axis
#define RS_KMAX 12  // largest number of outputs reading one input along an axis (backward)

__device__ __forceinline__ void tile_origin(const TileGrid& tg, int& b, int& x0, int& y0, int& z0) {
    int t = blockIdx.x;
    z0 = (t % tg.nz) * tg.tz; t /= tg.nz;
    y0 = (t % tg.ny) * tg.ty; t /= tg.ny;
    x0 = (t % tg.nx) * tg.tx;
    b = t / tg.nx;
}

// Per-axis interpolation tables of the tile live in LDS: they are computed once per workgroup by a
// few threads instead of once per lane (the index arithmetic, not the gather, was the bottleneck).
template <typename T>
__global__ void __launch_bounds__(256)
resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, AxisMap ax, AxisMap ay, AxisMap az, int C, TileGrid tg) {
    __shared__ int s_i0[3][RS_MAXT], s_i1[3][RS_MAXT];
    __shared__ float s_w[3][RS_MAXT];
    int b, ox0, oy0, oz0;
    tile_origin(tg, b, ox0, oy0, oz0);
    if (threadIdx.x < 3 * RS_MAXT) {
        const int a = threadIdx.x / RS_MAXT, k = threadIdx.x % RS_MAXT;
        const AxisMap& m = a == 0 ? ax : (a == 1 ? ay : az);
        const int o = min((a == 0 ? ox0 : (a == 1 ? oy0 : oz0)) + k, m.out - 1);
        int i0, i1;
        float w1;
        axis_taps(m, o, i0, i1, w1);
        s_i0[a][k] = i0; s_i1[a][k] = i1; s_w[a][k] = w1;
    }
    __syncthreads();
    const int L = C >> 3;
    const int per_pass = 256 / L;
    const int lc = (int)threadIdx.x % L;
    const int nvox = tg.tx * tg.ty * tg.tz;
    if ((int)threadIdx.x >= per_pass * L) return;
    for (int slot = (int)threadIdx.x / L; slot < nvox; slot += per_pass) {
        const int kz = slot & (tg.tz - 1), ky = (slot >> tg.sz) & (tg.ty - 1), kx = slot >> (tg.sz + tg.sy);
        const int ox = ox0 + kx, oy = oy0 + ky, oz = oz0 + kz;
        if (ox >= ax.out || oy >= ay.out || oz >= az.out) continue;
        const float wx = s_w[0][kx], wy = s_w[1][ky], wz = s_w[2][kz];
        const T* xb = x + ((int64_t)b * ax.in * ay.in * az.in) * C + lc * 8;
        const int xs[2] = {s_i0[0][kx], s_i1[0][kx]}, ys[2] = {s_i0[1][ky], s_i1[1][ky]}, zs[2] = {s_i0[2][kz], s_i1[2][kz]};
        const float wxs[2] = {1.0f - wx, wx}, wys[2] = {1.0f - wy, wy}, wzs[2] = {1.0f - wz, wz};
        Raw8<T> t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            t[k].load(xb + (((int64_t)xs[k >> 2] * ay.in + ys[(k >> 1) & 1]) * az.in + zs[k & 1]) * C);
        __builtin_amdgcn_sched_barrier(0);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // x outer, y, z inner
            const float w = wxs[k >> 2] * wys[(k >> 1) & 1] * wzs[k & 1];
            const Vec8<T> v = t[k].get();
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * v.v[j];
        }
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
        o.store(y + ((((int64_t)b * ax.out + ox) * ay.out + oy) * az.out + oz) * C + lc * 8);
    }
}

// ---- up-sampling, 2 x 2 x 2 outputs per lane ---------------------------------------------------------------------------
// When the grid grows along every axis (the U-Net's up path: 96 x 32 x 24 -> 192 x 64 x 48), the two outputs 2k and 2k + 1
// of an axis read taps inside one window of three inputs [base, base + 2], base = i0(2k) (host-checked per axis with the
// kernel's own float arithmetic, pair_window_ok).  A lane then produces a 2 x 2 x 2 block of outputs for its 8 channels from
// the 3 x 3 x 3 window -- 27 loads for 8 outputs instead of 64 -- interpolating separably (z, then y, then x).  Unused
// window entries carry weight 0 and a clamped address.  Worth 5-9 % at 192 x 64 x 48 x 64 (2.4 -> 2.6 TB/s of stores; a
// plain write-only stream reaches 4.8-6.5 TB/s, profiles/r11_hbm_write_probe.txt): with 216 VGPRs the kernel now waits on
// its three rounds of L2-latency loads at two waves per SIMD instead of on the L1 load rate; longer z runs per tile
// (2 x 8 x 16 ... 2 x 2 x 64) were measured and are no faster.
struct PairTaps {
    int base;
    float w[2][3];  // w[output of the pair][window entry]
};
__host__ __device__ __forceinline__ bool pair_taps(const AxisMap& a, int o_even, PairTaps& p) {
    int i0, i1;
    float w1;
    axis_taps(a, min(o_even, a.out - 1), i0, i1, w1);
    p.base = i0;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int s = 0; s < 3; ++s) p.w[q][s] = 0.f;
        const int o = min(o_even + q, a.out - 1);
        axis_taps(a, o, i0, i1, w1);
        const int d0 = i0 - p.base, d1 = i1 - p.base;
        ok = ok && d0 >= 0 && d0 <= 2 && d1 >= 0 && d1 <= 2;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s == d0) p.w[q][s] += 1.0f - w1;
            if (s == d1) p.w[q][s] += w1;
        }
    }
    return ok;
}
static bool pair_window_ok(const AxisMap& a) {
    if (a.out < a.in) return false;
    PairTaps p;
    for (int o = 0; o < a.out; o += 2)
        if (!pair_taps(a, o, p)) return false;
    return true;
}

// (256, 3): three workgroups per CU = three waves per SIMD, i.e. at most 168 VGPRs -- the fp16 instantiation came out at 169
// (two waves per SIMD: 185 -> 309 us on the level-0 up-sampling) without the bound
template <typename T>
__global__ void __launch_bounds__(256, 3)
resize_up_pairs_kernel(const T* __restrict__ x, T* __restrict__ y, AxisMap ax, AxisMap ay, AxisMap az, int C, TileGrid tg) {
    __shared__ PairTaps s_p[3][RS_MAXT / 2];
    int b, ox0, oy0, oz0;
    tile_origin(tg, b, ox0, oy0, oz0);  // tile extents are even (host-checked), so pairs never straddle tiles
    if (threadIdx.x < 3 * (RS_MAXT / 2)) {
        const int a = threadIdx.x / (RS_MAXT / 2), k = threadIdx.x % (RS_MAXT / 2);
        const AxisMap& m = a == 0 ? ax : (a == 1 ? ay : az);
        PairTaps p;
        pair_taps(m, (a == 0 ? ox0 : (a == 1 ? oy0 : oz0)) + 2 * k, p);
        s_p[a][k] = p;
    }
    __syncthreads();
    const int L = C >> 3;
    const int per_pass = 256 / L;
    const int lc = (int)threadIdx.x % L;
    const int px = tg.tx >> 1, py = tg.ty >> 1, pz = tg.tz >> 1;
    const int nblk = px * py * pz;
    if ((int)threadIdx.x >= per_pass * L) return;
    for (int slot = (int)threadIdx.x / L; slot < nblk; slot += per_pass) {
        const int kz = slot % pz, ky = (slot / pz) % py, kx = slot / (pz * py);
        const int ox = ox0 + 2 * kx, oy = oy0 + 2 * ky, oz = oz0 + 2 * kz;
        if (ox >= ax.out || oy >= ay.out || oz >= az.out) continue;
        const PairTaps tx = s_p[0][kx], ty = s_p[1][ky], tz = s_p[2][kz];
        const T* xb = x + ((int64_t)b * ax.in * ay.in * az.in) * C + lc * 8;
        int zi[3], yi[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            zi[s] = min(tz.base + s, az.in - 1);
            yi[s] = min(ty.base + s, ay.in - 1);
        }
        float acc[2][2][2][8];  // [x out][y out][z out][channel]
#pragma unroll
        for (int i = 0; i < 64; ++i) (&acc[0][0][0][0])[i] = 0.f;
#pragma unroll 1  // one x plane of the window at a time: 9 loads in flight, ~170 VGPRs instead of 256
        for (int sx = 0; sx < 3; ++sx) {
            const int xi = min(tx.base + sx, ax.in - 1);
            Raw8<T> t[3][3];
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sz = 0; sz < 3; ++sz) t[sy][sz].load(xb + (((int64_t)xi * ay.in + yi[sy]) * az.in + zi[sz]) * C);
            // along z: rows[sy][z out]
            float rz[3][2][8];
#pragma unroll
            for (int sy = 0; sy < 3; ++sy) {
                const Vec8<T> v0 = t[sy][0].get(), v1 = t[sy][1].get(), v2 = t[sy][2].get();
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        rz[sy][q][j] = tz.w[q][0] * v0.v[j] + tz.w[q][1] * v1.v[j] + tz.w[q][2] * v2.v[j];
            }
            // along y, then into the two x outputs
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                for (int qz = 0; qz < 2; ++qz)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float r = ty.w[qy][0] * rz[0][qz][j] + ty.w[qy][1] * rz[1][qz][j] + ty.w[qy][2] * rz[2][qz][j];
                        acc[0][qy][qz][j] += tx.w[0][sx] * r;
                        acc[1][qy][qz][j] += tx.w[1][sx] * r;
                    }
        }
#pragma unroll
        for (int qx = 0; qx < 2; ++qx)
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                for (int qz = 0; qz < 2; ++qz) {
                    if (ox + qx >= ax.out || oy + qy >= ay.out || oz + qz >= az.out) continue;
                    Vec8<T> o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.v[j] = acc[qx][qy][qz][j];
                    o.store(y + ((((int64_t)b * ax.out + ox + qx) * ay.out + oy + qy) * az.out + oz + qz) * C + lc * 8);
                }
    }
}

// range [lo, hi] of outputs that can read input index i (a superset; weights decide)
__host__ __device__ __forceinline__ void axis_range(const AxisMap& a, int i, int& lo, int& hi) {
    if (a.scale > 0.f) {
        lo = max(0, (int)floorf((float)(i - 1) / a.scale) - 1);
        hi = min(a.out - 1, (int)ceilf((float)(i + 1) / a.scale) + 1);
    } else {
        lo = 0; hi = a.out - 1;  // out == 1 or in == 1: every output reads input 0
    }
}
// weight with which output o reads input i (adjoint of axis_taps); 0 if it does not
__host__ __device__ __forceinline__ float axis_weight(const AxisMap& a, int o, int i) {
    int i0, i1;
    float w1;
    axis_taps(a, o, i0, i1, w1);
    float w = 0.f;
    if (i0 == i) w += 1.0f - w1;
    if (i1 == i) w += w1;
    return w;
}
// first output with a non-zero weight on input i and the number of outputs up to the last one
__host__ __device__ __forceinline__ void axis_span(const AxisMap& a, int i, int& first, int& cnt) {
    int lo, hi;
    axis_range(a, i, lo, hi);
    first = hi + 1;
    int last = lo - 1;
    for (int o = lo; o <= hi; ++o)
        if (axis_weight(a, o, i) != 0.f) {
            first = min(first, o);
            last = o;
        }
    cnt = last - first + 1;
    if (cnt < 0) cnt = 0;
}

// adjoint gather: dx[i] = sum over outputs o of w(o, i) dy[o].  Per axis and tile entry the LDS
// tables hold the first contributing output, their number (<= RS_KMAX, host-checked) and weights.
// K bounds the z candidates at compile time: per (x, y) candidate row the K z-loads go out
// together (clamped addresses, zero weights for the unused ones).
template <typename T, int K>
__global__ void __launch_bounds__(256)
resize_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx, AxisMap ax, AxisMap ay, AxisMap az,
                  int C, TileGrid tg) {
    __shared__ int s_first[3][RS_MAXT], s_cnt[3][RS_MAXT];
    __shared__ float s_w[3][RS_MAXT][RS_KMAX];
    int b, ix0, iy0, iz0;
    tile_origin(tg, b, ix0, iy0, iz0);
    if (threadIdx.x < 3 * RS_MAXT) {
        const int a = threadIdx.x / RS_MAXT, k = threadIdx.x % RS_MAXT;
        const AxisMap& m = a == 0 ? ax : (a == 1 ? ay : az);
        const int i = min((a == 0 ? ix0 : (a == 1 ? iy0 : iz0)) + k, m.in - 1);
        int f, c;
        axis_span(m, i, f, c);
        s_first[a][k] = f; s_cnt[a][k] = c;
        for (int j = 0; j < RS_KMAX; ++j) s_w[a][k][j] = j < c ? axis_weight(m, f + j, i) : 0.f;
    }
    __syncthreads();
    const int L = C >> 3;
    const int per_pass = 256 / L;
    const int lc = (int)threadIdx.x % L;
    const int nvox = tg.tx * tg.ty * tg.tz;
    if ((int)threadIdx.x >= per_pass * L) return;
    for (int slot = (int)threadIdx.x / L; slot < nvox; slot += per_pass) {
        const int kz = slot & (tg.tz - 1), ky = (slot >> tg.sz) & (tg.ty - 1), kx = slot >> (tg.sz + tg.sy);
        const int ix = ix0 + kx, iy = iy0 + ky, iz = iz0 + kz;
        if (ix >= ax.in || iy >= ay.in || iz >= az.in) continue;
        const int fx = s_first[0][kx], cx = s_cnt[0][kx], fy = s_first[1][ky], cy = s_cnt[1][ky], fz = s_first[2][kz];
        float wz[K];
        int oz[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            oz[k] = min(fz + k, az.out - 1);
            wz[k] = s_w[2][kz][k];
        }
        const T* gb = dy + ((int64_t)b * ax.out * ay.out * az.out) * C + lc * 8;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int a = 0; a < cx; ++a) {
            const float wx = s_w[0][kx][a];
            for (int bb = 0; bb < cy; ++bb) {
                const float wxy = wx * s_w[1][ky][bb];
                const T* row = gb + (((int64_t)(fx + a) * ay.out + (fy + bb)) * az.out) * C;
                Raw8<T> t[K];
#pragma unroll
                for (int k = 0; k < K; ++k) t[k].load(row + (int64_t)oz[k] * C);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float w = wxy * wz[k];
                    const Vec8<T> v = t[k].get();
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += w * v.v[j];
                }
            }
        }
        const int64_t di = ((((int64_t)b * ax.in + ix) * ay.in + iy) * az.in + iz) * C + lc * 8;
        if (add != nullptr) {  // second gradient of the resampled tensor (its use as a U-Net skip), fused add
            Vec8<T> a;
            a.load(add + di);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += a.v[j];
        }
        Vec8<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
        o.store(dx + di);
    }
}

// General fallback of the adjoint (any resampling factor): one lane per (input voxel, 8 channels),
// candidate outputs scanned over a superset range with the weights recomputed on the fly
template <typename T>
__global__ void __launch_bounds__(256)
resize_bwd_generic_kernel(const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx, AxisMap ax, AxisMap ay,
                          AxisMap az, int C, int64_t total) {
    const int L = C >> 3;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lc = (int)(i % L);
    int64_t v = i / L;
    const int iz = (int)(v % az.in); v /= az.in;
    const int iy = (int)(v % ay.in); v /= ay.in;
    const int ix = (int)(v % ax.in);
    const int b = (int)(v / ax.in);
    int x0, x1, y0, y1, z0, z1;
    axis_range(ax, ix, x0, x1);
    axis_range(ay, iy, y0, y1);
    axis_range(az, iz, z0, z1);
    const T* gb = dy + ((int64_t)b * ax.out * ay.out * az.out) * C + lc * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ox = x0; ox <= x1; ++ox) {
        const float wx = axis_weight(ax, ox, ix);
        if (wx == 0.f) continue;
        for (int oy = y0; oy <= y1; ++oy) {
            const float wxy = wx * axis_weight(ay, oy, iy);
            if (wxy == 0.f) continue;
            const T* row = gb + (((int64_t)ox * ay.out + oy) * az.out) * C;
            for (int oz = z0; oz <= z1; ++oz) {
                const float w = wxy * axis_weight(az, oz, iz);
                if (w == 0.f) continue;
                Vec8<T> t;
                t.load(row + (int64_t)oz * C);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += w * t.v[j];
            }
        }
    }
    const int64_t di = ((((int64_t)b * ax.in + ix) * ay.in + iy) * az.in + iz) * C + lc * 8;
    if (add != nullptr) {
        Vec8<T> a;
        a.load(add + di);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += a.v[j];
    }
    Vec8<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    o.store(dx + di);
}

static int resize_args_ok(int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C) {
    return B > 0 && Xi > 0 && Yi > 0 && Zi > 0 && Xo > 0 && Yo > 0 && Zo > 0 && C > 0;
}
static TileGrid make_tiles(int X, int Y, int Z, int C, int64_t samples = 0) {
    // ~8 passes of (256 / L) voxels: 256 voxels up to 64 channels, halved for every doubling beyond
    const int L = C >> 3;
    int vox = 256;
    while (vox > 8 && vox * L > 8 * 256) vox >>= 1;
    // samples > 0 (the adjoint): small tensors get smaller tiles, down to one pass per workgroup, until ~1024 workgroups
    // exist -- a tile's passes are chains of dependent gathers, and the 48x16x12 / 24x8x6 levels left 576 / 144 / 72
    // eight-pass workgroups on 256 CUs (100 / 68 / 30 us for 141 / 35 / 8 MB)
    const int one_pass = 256 / L > 8 ? 256 / L : 8;
    while (samples > 0 && vox > one_pass && samples * X * Y * Z / vox < 1024) vox >>= 1;
    TileGrid t;
    t.tz = 8; t.ty = 8; t.tx = 4;
    // shrink x first, then y, then z, down to the voxel budget
    while (t.tx * t.ty * t.tz > vox && t.tx > 1) t.tx >>= 1;
    while (t.tx * t.ty * t.tz > vox && t.ty > 1) t.ty >>= 1;
    while (t.tx * t.ty * t.tz > vox && t.tz > 1) t.tz >>= 1;
    t.sy = 0; while ((1 << t.sy) < t.ty) ++t.sy;
    t.sz = 0; while ((1 << t.sz) < t.tz) ++t.sz;
    t.nx = ceil_div(X, t.tx); t.ny = ceil_div(Y, t.ty); t.nz = ceil_div(Z, t.tz);
    return t;
}
extern "C" int tdx_resize_fwd(const void* x, void* y, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(x && y && resize_args_ok(B, Xi, Yi, Zi, Xo, Yo, Zo, C));
    if (C % 8 || C / 8 > 256) return TDX_ESHAPE;
    const TileGrid tg = make_tiles(Xo, Yo, Zo, C, B);
    const int64_t blocks = (int64_t)B * tg.nx * tg.ny * tg.nz;
    const AxisMap mx = make_axis(Xi, Xo), my = make_axis(Yi, Yo), mz = make_axis(Zi, Zo);
    // measured (tools/micro/resize_bench.py, B = 8): 64 channels 281 -> 256 us, 128 channels 65 -> 69 us: narrow rows only
    if (C <= 64 && !((tg.tx | tg.ty | tg.tz) & 1) && Xo > Xi && Yo > Yi && Zo > Zi && pair_window_ok(mx) &&
        pair_window_ok(my) && pair_window_ok(mz)) {
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_up_pairs_kernel<T>), dim3((unsigned)blocks), dim3(256), 0,
                                                      as_stream(stream), (const T*)x, (T*)y, mx, my, mz, C, tg));
        return tdx_launch_status();
    }
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_fwd_kernel<T>), dim3((unsigned)blocks), dim3(256), 0,
                                                  as_stream(stream), (const T*)x, (T*)y, make_axis(Xi, Xo),
                                                  make_axis(Yi, Yo), make_axis(Zi, Zo), C, tg));
    return tdx_launch_status();
}

// largest number of outputs that read one input along an axis (exact, same float arithmetic)
static int axis_max_span(const AxisMap& a) {
    int m = 0;
    for (int i = 0; i < a.in; ++i) {
        int f, c;
        axis_span(a, i, f, c);
        m = c > m ? c : m;
    }
    return m;
}

extern "C" int tdx_resize_bwd(const void* dy, const void* add, void* dx, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C,
                              int dtype, void* stream) {
    TDX_CHECK_ARG(dy && dx && resize_args_ok(B, Xi, Yi, Zi, Xo, Yo, Zo, C));
    if (C % 8 || C / 8 > 256) return TDX_ESHAPE;
    const AxisMap ax = make_axis(Xi, Xo), ay = make_axis(Yi, Yo), az = make_axis(Zi, Zo);
    const int kx = axis_max_span(ax), ky = axis_max_span(ay), kz = axis_max_span(az);
    if (kx > RS_KMAX || ky > RS_KMAX || kz > RS_KMAX) {  // more than ~6x upsampling along an axis: general kernel
        const int64_t total = (int64_t)B * Xi * Yi * Zi * (C / 8);
        TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_bwd_generic_kernel<T>), dim3(ceil_div(total, 256)), dim3(256), 0,
                                                      as_stream(stream), (const T*)dy, (const T*)add, (T*)dx, ax, ay, az, C, total));
        return tdx_launch_status();
    }
    const TileGrid tg = make_tiles(Xi, Yi, Zi, C, B);
    const dim3 grid((unsigned)((int64_t)B * tg.nx * tg.ny * tg.nz));
#define RS_BWD(KV)                                                                                                     \
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_bwd_kernel<T, KV>), grid, dim3(256), 0, as_stream(stream),     \
                                                  (const T*)dy, (const T*)add, (T*)dx, ax, ay, az, C, tg))
    if (kz <= 2) RS_BWD(2);
    else if (kz <= 4) RS_BWD(4);
    else if (kz <= 6) RS_BWD(6);
    else RS_BWD(RS_KMAX);
#undef RS_BWD
    return tdx_launch_status();
}
