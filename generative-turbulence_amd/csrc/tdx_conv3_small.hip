// bf16 MFMA 3x3x3 convolution for SMALL grids (the deep U-Net levels: 24 x 8 x 6 and 12 x 4 x 3 voxels, 256-1024
// channels), forward and data gradient.  gfx950.
//
// The brick kernel (tdx_conv3_mfma.hip) gives such a launch 96-288 workgroups, each filling 28-75 % of a 4 x 8 x 8
// brick and walking K = 27 x 512 alone on its CU, one 73-KB stage per 2.8 us: 130-600 TFLOP/s, and another 55-80 us
// per layer for the halo-shell kernel of the data gradient.  Here the GEMM is cut the other way:
//
//   M rows     = voxels of a "virtual grid", packed densely into 32-row M tiles regardless of brick shapes.  Forward:
//                the real grid, sources clamped (replicate padding).  Data gradient: the PADDED grid (X+2)(Y+2)(Z+2)
//                with zero sources outside the real grid -- the adjoint on the padded grid, whose rows the reduce pass
//                folds onto the voxels they clamp to; no separate shell launch.
//   workgroup  = (row group, 32 output channels, K split).  A row group is a few whole samples or an x slab of one,
//                at most 28 M tiles = 7 per wave; its sources live in LDS as a zero- / clamp-filled image with a
//                one-voxel rim, so a tap is a uniform offset and any row can sit in any lane.
//   K split    = contiguous ranges of 16-channel slices; every workgroup stores its fp32 partial tile to
//                slab[split][virtual voxel][channel], and conv3_small_reduce_kernel sums the splits (and, for the data
//                gradient, the up to 8 padded positions that clamp onto a voxel), adds bias / the residual addend,
//                converts and stores.  The split is what fills the chip: 12 x 4 x 3 x 6 samples is 27 M tiles in all.
//   staging    = LDS-DMA (global_load_lds_dwordx4), double-buffered: slice c + 1 lands while slice c's 27 taps run;
//                zero fill = DMA from a zeroed 16-B block.  No staging registers: 7 accumulator tiles + fragments fit.
//
// Slab traffic is the price (S x rows x N x 4 B written and read once); at these sizes it is 5-40 MB per layer.
#include "tdx_common.h"
#include "tdx_conv3_small_kernel.h"

// out[b, u, :] = sum over splits and over the virtual voxels v that map onto real voxel u of slab[split][b][v][:]
//                (+ bias) (+ addend), as T.  fold == 0: v = u (forward).  fold == 1: the virtual grid is the padded
//                grid and v ranges over the positions that clamp onto u (data gradient).  The result has N channels,
//                split over out1 (channels [0, D1)) and out2.
template <typename T>
__global__ void __launch_bounds__(256)
conv3_small_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias, T* __restrict__ out1, int D1,
                          T* __restrict__ out2, const T* __restrict__ add1, const T* __restrict__ add2, int B, int X,
                          int Y, int Z, int N, int nsplit, int fold) {
    const int groups = N >> 3;
    const int64_t total = (int64_t)B * X * Y * Z * groups;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % groups);
    int64_t u = idx / groups;
    const int uz = (int)(u % Z); int64_t t = u / Z;
    const int uy = (int)(t % Y); t /= Y;
    const int ux = (int)(t % X);
    const int b = (int)(t / X);
    const int E[3] = {X + 2 * fold, Y + 2 * fold, Z + 2 * fold}, c[3] = {ux, uy, uz}, R[3] = {X, Y, Z};
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fold ? (c[a] == 0 ? 0 : c[a] + 1) : c[a];
        hi[a] = fold ? (c[a] == R[a] - 1 ? R[a] + 1 : c[a] + 1) : c[a];
    }
    const int64_t Vv = (int64_t)E[0] * E[1] * E[2];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[cg * 8 + j] : 0.f;
    // (position, split) pairs in a fixed order, four pairs' loads in flight per trip: a rolled load / add loop is a chain of
    // memory round trips, and this pass runs on the tiny tensors of the deep levels, where latency is all it costs
    const int n0 = hi[0] - lo[0] + 1, n1 = hi[1] - lo[1] + 1, n2 = hi[2] - lo[2] + 1, npos = n0 * n1 * n2, total_pairs = npos * nsplit;
    auto pair_ptr = [&](int k) {
        const int s = k / npos, q = k - s * npos;
        const int v0 = lo[0] + q / (n1 * n2), v1 = lo[1] + (q / n2) % n1, v2 = lo[2] + q % n2;
        return slab + (((int64_t)s * B + b) * Vv + ((int64_t)v0 * E[1] + v1) * E[2] + v2) * N + cg * 8;
    };
    for (int k0 = 0; k0 < total_pairs; k0 += 4) {
        float4 a[4], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = pair_ptr(min(k0 + u, total_pairs - 1));
            a[u] = *reinterpret_cast<const float4*>(p);
            bq[u] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < total_pairs) {
                acc[0] += a[u].x; acc[1] += a[u].y; acc[2] += a[u].z; acc[3] += a[u].w;
                acc[4] += bq[u].x; acc[5] += bq[u].y; acc[6] += bq[u].z; acc[7] += bq[u].w;
            }
    }
    const int n = cg * 8;
    const int64_t vox = idx / groups;
    const bool first = n < D1;
    T* dst = first ? out1 + vox * D1 + n : out2 + vox * (N - D1) + (n - D1);
    const T* asrc = first ? (add1 ? add1 + vox * D1 + n : nullptr) : (add2 ? add2 + vox * (N - D1) + (n - D1) : nullptr);
    Vec8<T> o;
    if (asrc) {
        o.load(asrc);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] += acc[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = acc[j];
    }
    o.store(dst);
}

// ------------------------------------------------------------------------------------------------ host side
// Plan a launch; false if the small-grid kernel does not apply (grid too large, shapes, no arena, arena too small).
static bool small_plan(SmallGeom& g, int B, int X, int Y, int Z, int K, int N, bool data_gradient, bool split,
                       size_t arena_bytes, size_t& lds_bytes) {
    static const int mode = getenv("TDX_CONV3_SMALL") ? atoi(getenv("TDX_CONV3_SMALL")) : 1;
    if (mode == 0) return false;
    if ((K % SM_KC) || (N % SM_BN) || K < 128) return false;
    const int pad = data_gradient ? 1 : 0;
    g.B = B;
    g.Ev[0] = X + 2 * pad; g.Ev[1] = Y + 2 * pad; g.Ev[2] = Z + 2 * pad;
    g.Es[0] = X; g.Es[1] = Y; g.Es[2] = Z;
    g.off = pad; g.clamp = data_gradient ? 0 : 1;
    g.K = K; g.N = N;
    const int64_t rows_total = (int64_t)B * g.Ev[0] * g.Ev[1] * g.Ev[2];
    // Where it pays (tools/conv_bench.py, B = 6, us per launch, against brick kernel [+ halo-shell kernel]):
    //   12 x 4 x 3, 512 -> 512:    forward 33 vs 91;   data gradient 73 vs 141 + 55
    //   24 x 8 x 6, 512 -> 512:    forward 154 vs 162; 1024 -> 256: 149 vs 181; 256 -> 512: 90 vs 85
    //   24 x 8 x 6 data gradients: 183-353 vs 166-297 (the padded grid has 1.8x the rows there) -> brick kernels
    // TDX_CONV3_SMALL_ROWS moves the row gate (tests, A/B runs; read per call).
    const char* env_rows = getenv("TDX_CONV3_SMALL_ROWS");
    const int max_total = env_rows ? atoi(env_rows) : 6000;
    const bool deep_forward = !data_gradient && K >= 512 && rows_total <= 24000;
    if (rows_total > max_total && !(deep_forward && !env_rows)) return false;
    const int plane = g.Ev[1] * g.Ev[2], sample = g.Ev[0] * plane;
    const int max_rows = SM_MAX_TILES * 32;
    if (plane > max_rows) return false;
    if (sample <= max_rows) {  // whole samples per group; prefer >= 2 groups so that two row groups share the weights' L2 lines
        g.nbg = max(1, min(B, max_rows / sample));
        if (g.nbg == B && B > 1) g.nbg = (B + 1) / 2;
        g.xs = g.Ev[0]; g.gx = 1;
    } else {
        g.nbg = 1;
        g.xs = max_rows / plane;  // as many whole x planes as fit 28 M tiles (xs * plane <= max_rows; plane <= max_rows)
        g.gx = ceil_div(g.Ev[0], g.xs);
        g.xs = ceil_div(g.Ev[0], g.gx);  // even slabs
    }
    g.Ix = g.xs + 2; g.Iy = g.Ev[1] + 2; g.Iz = g.Ev[2] + 2;
    int img = g.Ix * g.Iy * g.Iz;
    auto lds_for = [&](int nbg) { return (size_t)4 * (((size_t)nbg * img + 63) & ~(size_t)63) * 16 + (size_t)4 * (27 * SM_BN * 16 + 512); };
    while (g.nbg > 1 && ((int64_t)g.nbg * img > 2048 || lds_for(g.nbg) > 160 * 1024)) --g.nbg;  // DMA plan: 4 x 8 x 64 entries
    if ((int64_t)g.nbg * img > 2048) return false;
    {
        // z stride of the image: a lane reads the fragment of ROW r of a densely packed M tile, i.e. of voxel r of a (y, z)-ordered
        // run, so the 16-B slots a ds_read_b128 touches are runs of Ev[2] entries one image row apart; with the natural stride
        // Ev[2] + 2 that costs 1.6-1.9x a conflict-free read (tools/micro/lds_pattern_probe.hip: 68-78 ticks against 41), with the
        // strides below 1.43x (59: the best any stride <= 16 reaches).  The padding entries are copied like rim entries and never
        // read.  Taken only where the group still fits the DMA plan and the LDS as planned (TDX_SMALL_ZPAD=0: off, A/B switch)
        static const int good[9] = {0, 0, 0, 9, 0, 9, 10, 0, 12};  // Ev[2] -> stride
        const char* env = getenv("TDX_SMALL_ZPAD");
        const int want = (g.Ev[2] <= 8 && !(env && atoi(env) == 0)) ? good[g.Ev[2]] : 0;
        if (want > g.Iz) {
            const int padded = g.Ix * g.Iy * want, keep = img;
            img = padded;
            if ((int64_t)g.nbg * padded <= 2048 && lds_for(g.nbg) <= 160 * 1024) g.Iz = want; else img = keep;
        }
    }
    lds_bytes = lds_for(g.nbg);
    if (lds_bytes > 160 * 1024) return false;
    // K splits.  One workgroup per CU (LDS), so a launch runs in rounds of 256 workgroups; a split count is judged by
    // rounds x slices per workgroup (at ~3.5 us per slice) plus the slab round trip it causes (written and read once
    // at ~5 TB/s), the smallest estimate wins: 12 x 4 x 3 forward 8 splits (1 round x 4 slices), 24 x 8 x 6 with 192
    // tile-groups 4 splits (3 full rounds x 8 slices instead of 2 rounds, the second half empty, x 16)
    const int ngroups = ceil_div(B, g.nbg) * g.gx, nslices = K / SM_KC;
    const int64_t base = (int64_t)ngroups * (N / SM_BN);
    int splits = 1;
    double best = 1e30;
    for (int sp = 1; sp <= 16 && sp * 2 <= std::max(nslices, 2); ++sp) {
        if ((size_t)sp * rows_total * N * 4 + 64 > arena_bytes) break;
        const double rounds = (double)ceil_div(base * sp, 256), per = (double)ceil_div(nslices, sp);
        const double est = rounds * per * (split ? 9e-6 : 3.5e-6) + (sp > 1 ? 2.0 * sp * rows_total * N * 4 / 5e12 : rows_total * N * 8.0 / 5e12);
        if (est < best * 0.97) { best = est; splits = sp; }
    }
    if ((size_t)splits * rows_total * N * 4 + 64 > arena_bytes) return false;
    g.per_split = ceil_div(nslices, splits);
    g.nsplit = ceil_div(nslices, g.per_split);
    return true;
}

int conv3_small_go_3b(SMALL_GO_ARGS);
int conv3_small_go_3s(SMALL_GO_ARGS);
int conv3_small_go_4b(SMALL_GO_ARGS);
int conv3_small_go_4s(SMALL_GO_ARGS);
int conv3_small_go_5b(SMALL_GO_ARGS);
int conv3_small_go_5s(SMALL_GO_ARGS);
int conv3_small_go_6b(SMALL_GO_ARGS);
int conv3_small_go_6s(SMALL_GO_ARGS);
int conv3_small_go_7b(SMALL_GO_ARGS);
int conv3_small_go_7s(SMALL_GO_ARGS);
int conv3_small_go_3h(SMALL_GO_ARGS);
int conv3_small_go_4h(SMALL_GO_ARGS);
int conv3_small_go_5h(SMALL_GO_ARGS);
int conv3_small_go_6h(SMALL_GO_ARGS);
int conv3_small_go_7h(SMALL_GO_ARGS);

// FMT: 0 bf16 tensors, 1 split-precision fp32 tensors, 2 fp16 tensors
template <int FMT>
static int small_dispatch(int mtw, SMALL_GO_ARGS) {
#define GO(M) (FMT == 1 ? conv3_small_go_##M##s(x1, C1, x2, C2, wp, slab, zero16, g, lds, lo_offset, st)   \
               : FMT == 2 ? conv3_small_go_##M##h(x1, C1, x2, C2, wp, slab, zero16, g, lds, lo_offset, st) \
                          : conv3_small_go_##M##b(x1, C1, x2, C2, wp, slab, zero16, g, lds, lo_offset, st))
    switch (mtw) {
        case 1: case 2: case 3: return GO(3);
        case 4: return GO(4);
        case 5: return GO(5);
        case 6: return GO(6);
        case 7: return GO(7);
        default: return TDX_ESHAPE;  // more than 28 M tiles per row group: not a launch small_plan produces
    }
#undef GO
}

// would conv3_small_launch take this call? (bookkeeping: tdx_conv3_fwd_kernel)
bool conv3_small_applies(int C1, int C2, int B, int X, int Y, int Z, int N, bool data_gradient, bool split) {
    if (tdx_scratch_ptr() == nullptr || (C1 % SM_KC) || (C2 % SM_KC)) return false;
    SmallGeom g;
    size_t lds = 0;
    if (!small_plan(g, B, X, Y, Z, C1 + C2, N, data_gradient, split, tdx_scratch_bytes(), lds)) return false;
    return ceil_div(ceil_div(g.nbg * g.xs * g.Ev[1] * g.Ev[2], 32), 4) <= SM_MAX_TILES / 4;
}

// Forward (data_gradient == false: y = conv3([x1 | x2]) + bias, N = Cout) or data gradient (x1 = dy with K = C1
// channels, x2 unused; result N channels split over out1 [0, D1) / out2, plus addends).  split == false: bf16 tensors,
// wp = the bf16 packed weight; split == true: fp32 tensors, wp = the split-precision packed weight (hi image, lo image).
// Returns TDX_ESHAPE when the launch is not a small-grid case (the caller then takes the brick kernels).
int conv3_small_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* out1, int D1,
                       void* out2, const void* add1, const void* add2, int B, int X, int Y, int Z, int N, bool data_gradient,
                       bool split, hipStream_t st, bool hf) {
    char* arena = (char*)tdx_scratch_ptr();
    if (arena == nullptr) return TDX_ESHAPE;
    if ((C1 % SM_KC) || (C2 % SM_KC)) return TDX_ESHAPE;
    SmallGeom g;
    size_t lds = 0;
    if (!small_plan(g, B, X, Y, Z, C1 + C2, N, data_gradient, split, tdx_scratch_bytes(), lds)) return TDX_ESHAPE;
    float* slab = reinterpret_cast<float*>(arena + 64);
    const int rows = g.nbg * g.xs * g.Ev[1] * g.Ev[2];
    const int mtw = ceil_div(ceil_div(rows, 32), 4);
    if (mtw > SM_MAX_TILES / 4) return TDX_ESHAPE;  // never drop rows silently: the brick kernels take the call
    const int64_t lo = (int64_t)27 * (C1 + C2) * N;  // elements between the hi and the lo weight image
    if (split && hf) return TDX_EINVAL;
    const int rc = split ? small_dispatch<1>(mtw, x1, C1, x2, C2, wp, slab, arena, g, lds, lo, st)
                   : hf ? small_dispatch<2>(mtw, x1, C1, x2, C2, wp, slab, arena, g, lds, 0, st)
                         : small_dispatch<0>(mtw, x1, C1, x2, C2, wp, slab, arena, g, lds, 0, st);
    if (rc != TDX_OK) return rc;
    const int64_t total = (int64_t)B * X * Y * Z * (N / 8);
    if (split)
        hipLaunchKernelGGL(conv3_small_reduce_kernel<float>, dim3(ceil_div(total, 256)), dim3(256), 0, st, slab, bias, (float*)out1,
                           D1, (float*)out2, (const float*)add1, (const float*)add2, B, X, Y, Z, N, g.nsplit, data_gradient ? 1 : 0);
    else if (hf)
        hipLaunchKernelGGL(conv3_small_reduce_kernel<f16>, dim3(ceil_div(total, 256)), dim3(256), 0, st, slab, bias, (f16*)out1,
                           D1, (f16*)out2, (const f16*)add1, (const f16*)add2, B, X, Y, Z, N, g.nsplit, data_gradient ? 1 : 0);
    else
        hipLaunchKernelGGL(conv3_small_reduce_kernel<bf16>, dim3(ceil_div(total, 256)), dim3(256), 0, st, slab, bias, (bf16*)out1,
                           D1, (bf16*)out2, (const bf16*)add1, (const bf16*)add2, B, X, Y, Z, N, g.nsplit, data_gradient ? 1 : 0);
    return tdx_launch_status();
}
