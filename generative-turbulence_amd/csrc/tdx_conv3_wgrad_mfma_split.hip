// Split-precision ("bf16x2") MFMA weight gradient of the replicate-padded 3x3x3 convolution for fp32 tensors
// (gfx950, opt-in with TDX_CONV_SPLIT; see tdx_conv3_mfma_split.hip for the arithmetic):
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]
//                  ~= sum_v  xh * dyh + xl * dyh + xh * dyl        (bf16 hi / lo terms, fp32 accumulation)
//
// The structure is that of the bf16 kernel (tdx_conv3_wgrad_mfma.hip): a TN GEMM per tap over the voxels of
// 4 x 8 x 8 bricks, both operands voxel-major, fragments by transposed LDS reads (ds_read_b64_tr_b16) of
// 64-B rows; a workgroup owns a 32 (ci) x 32 (co) tile of all 27 taps, wave w the taps w, w + 4, ...
// (7 x 16 accumulator registers).  Differences: the operands are loaded as fp32 and split into a hi and a lo
// bf16 image while they are written to LDS (x: 2 x 38 KB, dy: 2 x 16 KB -> one workgroup per CU), each
// (tap, K-step) issues three MFMAs, the output tile is 32 channels wide (NT = 1: the second register set of
// the double-buffered fragments takes the room of the second accumulator tile), and the bias gradient is
// summed from the fp32 staging registers, i.e. exactly.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define WS_BX 4
#define WS_BY 8
#define WS_BZ 8
#define WS_HY 10
#define WS_HZ 10
#define WS_NVOX (WS_BX * WS_BY * WS_BZ)              // 256
#define WS_NSTEPS (WS_NVOX / 16)                     // K-steps of 16 voxels
#define WS_NHALO ((WS_BX + 2) * WS_HY * WS_HZ)        // 600
#define WS_XBYTES (WS_NHALO * 64)                    // one image of the halo brick (32 bf16 channels per voxel)
#define WS_GBYTES (WS_NVOX * 64)                     // one image of the dy brick
#define WS_TAPS 7

struct WgradViewS {
    int B;
    int E[3], s[3], ws[3], nb[3];
    int batch;
};

bool conv3_wgrad_mfma_split_supported(int C1, int C2, int Cout) {
    const bool c1_ok = (C1 % 32) == 0 || (C2 == 0 && (C1 % 8) == 0);
    return C1 > 0 && c1_ok && (C2 % 32) == 0 && (Cout % 32) == 0;
}

__device__ __forceinline__ bf16x8 tr_frag_s(const unsigned char* base_lo, const unsigned char* base_hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ void split8w(const float4& a, const float4& b, uint4& hi, uint4& lo) {
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

__global__ void __launch_bounds__(256, 1)
conv3_wgrad_mfma_split_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                              const float* __restrict__ dy, float* __restrict__ dwp, float* __restrict__ dbias, WgradViewS gv,
                              int Cout, int nsplit, int n_ci_tiles, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sXh = smem;
    unsigned char* sXl = smem + WS_XBYTES;
    unsigned char* sGh = smem + 2 * WS_XBYTES;
    unsigned char* sGl = sGh + WS_GBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32;
    const int co0 = (tile / n_ci_tiles) * 32;
    const float* xs;
    int Cs, cbase;
    if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }

    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    // fragment lane geometry (as the bf16 kernel)
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;
    const int kh = g >> 1;

    f32x16 acc[WS_TAPS];
#pragma unroll
    for (int t = 0; t < WS_TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const bool do_bias = dbias != nullptr && ci0 == 0;
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    int a_off[WS_TAPS];  // byte offset inside an x image of this lane's fragment at step 0, per tap
#pragma unroll
    for (int t = 0; t < WS_TAPS; ++t) {
        const int tap = min(wave + 4 * t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        const int toff = (ex * WS_HY + ey) * WS_HZ + ez;
        a_off[t] = ((WS_HY + kh + 1) * WS_HZ + (q + 1) + toff) * 64 + col_off;
    }

    constexpr int XP = (WS_NHALO * 4 + 255) / 256;  // pieces of 8 fp32 channels per thread (10)
    constexpr int GP = (WS_NVOX * 4) / 256;          // 4
    float4 xreg[XP][2], greg[GP][2];

    auto load_brick = [&](int brick) {
        int bb = brick;
        const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
        const int by = bb % gv.nb[1]; bb /= gv.nb[1];
        const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
        const int b = bb;
        const int ox0 = bx * WS_BX, oy0 = by * WS_BY, oz0 = bz * WS_BZ;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            xreg[i][0] = xreg[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pc < WS_NHALO * 4 && cbase + (pc & 3) * 8 < Cs) {
                const int hv = pc >> 2, q4 = pc & 3;
                const int hx = hv / (WS_HY * WS_HZ), rem = hv - hx * (WS_HY * WS_HZ);
                const int hy = rem / WS_HZ, hz = rem - hy * WS_HZ;
                const int sx = min(max(ox0 + hx - 1, 0), gv.E[0] - 1), sy = min(max(oy0 + hy - 1, 0), gv.E[1] - 1),
                          sz = min(max(oz0 + hz - 1, 0), gv.E[2] - 1);
                const int64_t vox = (int64_t)b * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
                const float4* src = reinterpret_cast<const float4*>(xs + vox * Cs + cbase + q4 * 8);
                xreg[i][0] = src[0];
                xreg[i][1] = src[1];
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            const int v = pc >> 2, q8 = pc & 3;
            const int vx = ox0 + (v >> 6), vy = oy0 + ((v >> 3) & 7), vz = oz0 + (v & 7);
            greg[i][0] = greg[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2]) {
                const int64_t vox = (int64_t)b * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
                const float4* src = reinterpret_cast<const float4*>(dy + vox * Cout + co0 + q8 * 8);
                greg[i][0] = src[0];
                greg[i][1] = src[1];
            }
        }
    };
    auto store_brick = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            if (pc < WS_NHALO * 4) {
                uint4 hi, lo;
                split8w(xreg[i][0], xreg[i][1], hi, lo);
                *reinterpret_cast<uint4*>(sXh + pc * 16) = hi;
                *reinterpret_cast<uint4*>(sXl + pc * 16) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            uint4 hi, lo;
            split8w(greg[i][0], greg[i][1], hi, lo);
            *reinterpret_cast<uint4*>(sGh + pc * 16) = hi;
            *reinterpret_cast<uint4*>(sGl + pc * 16) = lo;
            if (do_bias) {
                bs[0] += greg[i][0].x; bs[1] += greg[i][0].y; bs[2] += greg[i][0].z; bs[3] += greg[i][0].w;
                bs[4] += greg[i][1].x; bs[5] += greg[i][1].y; bs[6] += greg[i][1].z; bs[7] += greg[i][1].w;
            }
        }
    };

    auto step_off = [&](int s) { return ((s >> 2) * WS_HY + 2 * (s & 3)) * WS_HZ * 64; };
    auto read_a = [&](int soff, int t, bf16x8& h, bf16x8& l) {
        const int o = a_off[t] + soff;
        h = tr_frag_s(sXh + o, sXh + o + 4 * 64);
        l = tr_frag_s(sXl + o, sXl + o + 4 * 64);
    };
    auto read_b = [&](int s, bf16x8& h, bf16x8& l) {
        const int o = (16 * s + 8 * kh + q) * 64 + col_off;
        h = tr_frag_s(sGh + o, sGh + o + 4 * 64);
        l = tr_frag_s(sGl + o, sGl + o + 4 * 64);
    };

    int brick = split;
    if (brick < nbricks) load_brick(brick);
    for (; brick < nbricks; brick += nsplit) {
        __syncthreads();
        store_brick();
        __syncthreads();
        if (brick + nsplit < nbricks) load_brick(brick + nsplit);  // in flight during the MFMA phase

        bf16x8 A0h[WS_TAPS], A0l[WS_TAPS], A1h[WS_TAPS], A1l[WS_TAPS], B0h, B0l, B1h, B1l;
#pragma unroll
        for (int t = 0; t < WS_TAPS; ++t) read_a(0, t, A0h[t], A0l[t]);
        read_b(0, B0h, B0l);
#pragma unroll 1
        for (int s2 = 0; s2 < WS_NSTEPS / 2; ++s2) {
            const int so = 2 * s2 + 1, sn = min(2 * s2 + 2, WS_NSTEPS - 1);
            const int off_o = step_off(so), off_n = step_off(sn);
            // term-major MFMA order: one wave per SIMD, so an MFMA that accumulates onto the result of the MFMA
            // right before it stalls ~12 cycles (tools/micro/mfma_peak: 1785 vs 2437 TFLOP/s); the three terms of a
            // tap are issued 7 MFMAs apart.  The next step's fragments are fetched during the first term.
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) {  // even step: compute set 0, fetch set 1
                read_a(off_o, t, A1h[t], A1l[t]);
                if (t == 0) read_b(so, B1h, B1l);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h[t], B0h, acc[t], 0, 0, 0);
                if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0l[t], B0h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0h[t], B0l, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * WS_TAPS, 0);
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) {  // odd step: compute set 1, fetch set 0
                read_a(off_n, t, A0h[t], A0l[t]);
                if (t == 0) read_b(sn, B0h, B0l);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1h[t], B1h, acc[t], 0, 0, 0);
                if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1l[t], B1h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < WS_TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1h[t], B1l, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * WS_TAPS, 0);
        }
    }

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < WS_TAPS; ++t) {
        const int ltap = wave + 4 * t;
        if (ltap < 27) {
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (ci >= Cin) continue;
                float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + r];
                if (slab_stride) dst[(int64_t)split * slab_stride] = acc[t][i];
                else atomicAdd(dst, acc[t][i]);
            }
        }
    }
    if (do_bias) {
        // threads with equal tid % 4 hold partial sums of the same 8 channels
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // [256][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 32) {
            const int q8 = tid >> 3, e = tid & 7;
            float t = 0.f;
            for (int k = q8; k < 256; k += 4) t += red[k * 8 + e];
            atomicAdd(&dbias[co0 + tid], t);
        }
    }
}

int conv3_wgrad_mfma_split_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias,
                                  int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs,
                                  int* nslab_out) {
    {   // grids that give every workgroup several bricks: the producer / consumer form
        const int rs = conv3_wgrad_split_ring_launch(x1, C1, x2, C2, dy, dwp, dbias, B, X, Y, Z, Cout, st, slabs, max_slabs, nslab_out);
        if (rs != TDX_ESHAPE) return rs;
    }
    const int Cin = C1 + C2;
    const int E[3] = {X, Y, Z}, gs[3] = {Y * Z, Z, 1}, gw[3] = {9, 3, 1};
    const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
    int best = 0;
    int64_t best_n = -1;
    for (int c = 0; c < 3; ++c) {
        const int64_t n = (int64_t)ceil_div(E[cand[c][0]], WS_BX) * ceil_div(E[cand[c][1]], WS_BY) * ceil_div(E[cand[c][2]], WS_BZ);
        if (best_n < 0 || n < best_n) { best_n = n; best = c; }
    }
    WgradViewS g;
    g.B = B; g.batch = X * Y * Z;
    const int bdim[3] = {WS_BX, WS_BY, WS_BZ};
    for (int k = 0; k < 3; ++k) {
        const int a = cand[best][k];
        g.E[k] = E[a]; g.s[k] = gs[a]; g.ws[k] = gw[a]; g.nb[k] = ceil_div(E[a], bdim[k]);
    }
    const int nbricks = B * g.nb[0] * g.nb[1] * g.nb[2];
    const int n_ci = (Cin + 31) / 32, n_co = Cout / 32;
    const int ntiles = n_ci * n_co;
    int nsplit = (256 + ntiles - 1) / ntiles;  // one workgroup per CU: one resident wave of workgroups
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    const size_t lds = (size_t)2 * WS_XBYTES + 2 * WS_GBYTES;
    dim3 grid((unsigned)(ntiles * nsplit));
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
    auto kern = conv3_wgrad_mfma_split_kernel;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)x1, C1, (const float*)x2, C2, (const float*)dy, out, dbias,
                       g, Cout, nsplit, n_ci, slab_stride);
    return tdx_launch_status();
}
