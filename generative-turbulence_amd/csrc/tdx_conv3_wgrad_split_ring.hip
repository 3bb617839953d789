// Split-precision ("bf16x2") MFMA weight gradient of the replicate-padded 3x3x3 convolution for fp32 tensors, producer /
// consumer form (gfx950): 8 computing + 4 loader waves per workgroup, one workgroup per CU.
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]  ~=  sum_v  xh * dyh + xl * dyh + xh * dyl
//
// The single-role kernel (tdx_conv3_wgrad_mfma_split.hip) lets one wave per SIMD load, split, store and multiply in turn:
// its matrix pipe is busy 52 % of the cycles, and with the re-staging removed 73 % (profiles/r12_split_staging_ablation.txt,
// section 5).  This kernel is tdx_conv3_wgrad_ring.hip's 32-wide form with split operands:
//   * a workgroup owns a 32 (ci) x 32 (co) tile of all 27 taps and walks 2 x 8 x 8 bricks (half the single-role kernel's:
//     hi + lo images of the halo'd x brick and of the dy brick are 67 KB, so that TWO sets fit the LDS);
//   * the 4 LOADER waves stage brick i + 1 into the other set while brick i is multiplied: global -> registers -> hi / lo
//     (bf16(v), bf16(v - hi)) -> LDS, zero rows for voxels beyond a ragged grid and channels beyond a half-filled ci tile
//     written from registers (no zero source needed); ONE barrier per brick;
//   * the 8 COMPUTING waves own 4 / 4 / 4 / 3 / 3 / 3 / 3 / 3 taps (7 x 3 MFMAs per SIMD and K step of 16 voxels); the three
//     terms of a tap are issued back to back on its accumulator (the SIMD's other computing wave fills the dependent-issue
//     stall), then the tap's x fragments are re-read for the next step; wave 7's spare slot multiplies ones with dy_hi and
//     dy_lo: the bias gradient.
// Same contract as conv3_wgrad_mfma_split_launch; TDX_ESHAPE = not a case for it (too few bricks per workgroup).
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define SR_BX 2
#define SR_BY 8
#define SR_BZ 8
#define SR_HY 10
#define SR_HZ 10
#define SR_NVOX (SR_BX * SR_BY * SR_BZ)                // 128
#define SR_NSTEPS (SR_NVOX / 16)                       // 8 K steps of 16 voxels
#define SR_NHALO ((SR_BX + 2) * SR_HY * SR_HZ)          // 400 halo'd voxels, 64-B rows (32 bf16 channels)
#define SR_XBYTES (SR_NHALO * 64)                      // one image (hi or lo) of the x brick
#define SR_GBYTES (SR_NVOX * 64)                       // one image of the dy brick
#define SR_SET (2 * SR_XBYTES + 2 * SR_GBYTES)         // 67 584 B
#define SR_CW 8                                        // computing waves
#define SR_LT 256                                      // loader threads

struct WgradSplitRingView {
    int B;
    int E[3];     // extents in the kernel's local axes (brick 2 x 8 x 8)
    int s[3];     // voxel strides
    int ws[3];    // weight-tap strides: global tap = sum_k (e_k + 1) * ws[k]
    int nb[3];    // bricks per axis
    int batch;    // voxels per sample
};

__device__ __forceinline__ bf16x8 sr_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ void sr_split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
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

__device__ __forceinline__ void sr_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(768, 3)
conv3_wgrad_split_ring_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                              const float* __restrict__ dy, float* __restrict__ dwp, float* __restrict__ dbias,
                              WgradSplitRingView gv, int Cout, int nsplit, int n_ci_tiles, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 sets][x hi | x lo | dy hi | dy lo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32, co0 = (tile / n_ci_tiles) * 32;
    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    if (wave >= SR_CW) {
        // =========================================================== loader waves
        const int lt = tid - SR_CW * 64;
        const float* xs;
        int Cs, cbase;
        if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }
        constexpr int XP = (SR_NHALO * 4 + SR_LT - 1) / SR_LT;  // pieces (halo voxel, 8 fp32 channels) per thread: 7
        constexpr int GP = (SR_NVOX * 4) / SR_LT;                // 2
        int xh[XP];  // hx | hy << 8 | hz << 16 | channels exist << 30
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = lt + i * SR_LT;
            const int hv = min(pc >> 2, SR_NHALO - 1), q4 = pc & 3;
            const int hx = hv / (SR_HY * SR_HZ), rem = hv - hx * (SR_HY * SR_HZ);
            const int hy = rem / SR_HZ, hz = rem - hy * SR_HZ;
            xh[i] = hx | (hy << 8) | (hz << 16) | ((cbase + q4 * 8 < Cs) ? (1 << 30) : 0);
        }
        auto stage = [&](int brick, int set) {
            int bb = brick;
            const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
            const int by = bb % gv.nb[1]; bb /= gv.nb[1];
            const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
            float4 xr[XP][2], gr[GP][2];
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int pc = lt + i * SR_LT;
                xr[i][0] = xr[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pc < SR_NHALO * 4 && ((xh[i] >> 30) & 1)) {
                    const int sx = min(max(bx * SR_BX + (xh[i] & 0xff) - 1, 0), gv.E[0] - 1);
                    const int sy = min(max(by * SR_BY + ((xh[i] >> 8) & 0xff) - 1, 0), gv.E[1] - 1);
                    const int sz = min(max(bz * SR_BZ + ((xh[i] >> 16) & 0xff) - 1, 0), gv.E[2] - 1);
                    const int64_t vox = (int64_t)bb * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
                    const float4* src = reinterpret_cast<const float4*>(xs + vox * Cs + cbase + (pc & 3) * 8);
                    xr[i][0] = src[0];
                    xr[i][1] = src[1];
                }
            }
#pragma unroll
            for (int i = 0; i < GP; ++i) {
                const int pc = lt + i * SR_LT;
                const int v = pc >> 2, q8 = pc & 3;
                const int vx = bx * SR_BX + (v >> 6), vy = by * SR_BY + ((v >> 3) & 7), vz = bz * SR_BZ + (v & 7);
                gr[i][0] = gr[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2]) {
                    const int64_t vox = (int64_t)bb * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
                    const float4* src = reinterpret_cast<const float4*>(dy + vox * Cout + co0 + q8 * 8);
                    gr[i][0] = src[0];
                    gr[i][1] = src[1];
                }
            }
            unsigned char* sX = smem + set * SR_SET;
            unsigned char* sG = sX + 2 * SR_XBYTES;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int pc = lt + i * SR_LT;
                if (pc < SR_NHALO * 4) {
                    uint4 hi, lo;
                    sr_split8(xr[i][0], xr[i][1], hi, lo);
                    *reinterpret_cast<uint4*>(sX + pc * 16) = hi;
                    *reinterpret_cast<uint4*>(sX + SR_XBYTES + pc * 16) = lo;
                }
            }
#pragma unroll
            for (int i = 0; i < GP; ++i) {
                const int pc = lt + i * SR_LT;
                uint4 hi, lo;
                sr_split8(gr[i][0], gr[i][1], hi, lo);
                *reinterpret_cast<uint4*>(sG + pc * 16) = hi;
                *reinterpret_cast<uint4*>(sG + SR_GBYTES + pc * 16) = lo;
            }
        };
        int brick = split, it = 0;
        if (brick < nbricks) stage(brick, 0);
        for (; brick < nbricks; brick += nsplit, ++it) {
            sr_barrier();  // brick `it` is staged, the computing waves are done with brick it - 1
            if (brick + nsplit < nbricks) stage(brick + nsplit, (it + 1) & 1);
        }
        return;
    }

    // =============================================================== computing waves
    // waves 0-2 own 4 taps, waves 3-7 own 3 (27 = 3 x 4 + 5 x 3); wave 7's fourth slot sums the bias gradient
    const int first = wave < 3 ? 4 * wave : 12 + 3 * (wave - 3);
    const int cnt = wave < 3 ? 4 : 3;
    const bool bias_slot = wave == 7;

    // fragment lane geometry (tdx_conv3_wgrad_mfma.hip): a K step is 16 voxels; lane group g of 16 lanes reads voxel rows
    // 8 kh + q and + 4, columns 16 (g & 1) + 4 p .. + 3
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;
    const int kh = g >> 1;
    int a_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = min(first + t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        a_off[t] = ((SR_HY + kh + 1) * SR_HZ + (q + 1) + (ex * SR_HY + ey) * SR_HZ + ez) * 64 + col_off;
    }
    const int b_row = (8 * kh + q) * 64 + col_off;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // specialised at compile time on whether the wave's fourth slot is in use; the K-step loop stays rolled, two steps per
    // trip (dy fragment sets by step parity)
    auto run = [&](auto fourth_c) {
        constexpr bool FOURTH = decltype(fourth_c)::value;
        int it = 0;
        for (int brick = split; brick < nbricks; brick += nsplit, ++it) {
            sr_barrier();
            const unsigned char* bX = smem + (it & 1) * SR_SET;
            const unsigned char* bG = bX + 2 * SR_XBYTES + b_row;
            auto step_off = [&](int s) { return ((s >> 2) * SR_HY + 2 * (s & 3)) * SR_HZ * 64; };
            auto read_a = [&](int s, int t, bf16x8& h, bf16x8& l) {
                const unsigned char* ap = bX + a_off[t] + step_off(s);
                h = sr_tr_frag(ap, ap + 4 * 64);
                l = sr_tr_frag(ap + SR_XBYTES, ap + SR_XBYTES + 4 * 64);
            };
            auto read_b = [&](int s, bf16x8& h, bf16x8& l) {
                const unsigned char* bp = bG + s * (16 * 64);
                h = sr_tr_frag(bp, bp + 4 * 64);
                l = sr_tr_frag(bp + SR_GBYTES, bp + SR_GBYTES + 4 * 64);
            };
            bf16x8 Ah[4], Al[4], Bh[2], Bl[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) read_a(0, t, Ah[t], Al[t]);
            read_b(0, Bh[0], Bl[0]);

            auto step = [&](int s, int cur) {
                const int sn = min(s + 1, SR_NSTEPS - 1), nx = cur ^ 1;
                read_b(sn, Bh[nx], Bl[nx]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bh[cur], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[t], Bh[cur], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[t], Bl[cur], acc[t], 0, 0, 0);
                    read_a(sn, t, Ah[t], Al[t]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
                if (FOURTH) {  // the fourth tap (waves 0-2) or the bias column sums (wave 7: ones x (dy_hi + dy_lo))
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bias_slot ? ones : Ah[3], Bh[cur], acc[3], 0, 0, 0);
                    if (!bias_slot) acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[3], Bh[cur], acc[3], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bias_slot ? ones : Ah[3], Bl[cur], acc[3], 0, 0, 0);
                    if (!bias_slot) read_a(sn, 3, Ah[3], Al[3]);
                }
            };
#pragma unroll 1
            for (int s2 = 0; s2 < SR_NSTEPS / 2; ++s2) {
                step(2 * s2, 0);
                step(2 * s2 + 1, 1);
            }
        }
    };
    if (cnt == 4 || bias_slot) run(std::true_type{}); else run(std::false_type{});

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ltap = first + i;
        if (i < cnt) {
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (ci >= Cin) continue;  // half-filled last tile
                float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + r];
                if (slab_stride) dst[(int64_t)split * slab_stride] = acc[i][e];
                else atomicAdd(dst, acc[i][e]);
            }
        } else if (bias_slot && i == 3 && dbias != nullptr && ci0 == 0 && hh == 0) {
            atomicAdd(&dbias[co0 + r], acc[i][0]);
        }
    }
}

int conv3_wgrad_split_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B,
                                  int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out) {
    {
        const char* env = getenv("TDX_WGRAD_SPLIT_RING");  // A/B switch, read per call: 0 = off
        if (env && atoi(env) == 0) return TDX_ESHAPE;
    }
    if (!conv3_wgrad_mfma_split_supported(C1, C2, Cout)) return TDX_ESHAPE;
    const int Cin = C1 + C2;
    // local axes: brick 2 x 8 x 8; the short axis goes where it leaves the fewest bricks
    const int E[3] = {X, Y, Z}, gs[3] = {Y * Z, Z, 1}, gw[3] = {9, 3, 1};
    const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
    int best = 0;
    int64_t best_n = -1;
    for (int c = 0; c < 3; ++c) {
        const int64_t n = (int64_t)ceil_div(E[cand[c][0]], SR_BX) * ceil_div(E[cand[c][1]], SR_BY) * ceil_div(E[cand[c][2]], SR_BZ);
        if (best_n < 0 || n < best_n) { best_n = n; best = c; }
    }
    WgradSplitRingView g;
    g.B = B; g.batch = X * Y * Z;
    const int bdim[3] = {SR_BX, SR_BY, SR_BZ};
    for (int k = 0; k < 3; ++k) {
        const int a = cand[best][k];
        g.E[k] = E[a]; g.s[k] = gs[a]; g.ws[k] = gw[a]; g.nb[k] = ceil_div(E[a], bdim[k]);
    }
    const int nbricks = B * g.nb[0] * g.nb[1] * g.nb[2];
    const int n_ci = (Cin + 31) / 32, n_co = Cout / 32;
    const int ntiles = n_ci * n_co;
    const int cus = tdx_persistent_cus();
    int nsplit = cus >= 256 ? (256 + ntiles - 1) / ntiles : std::max(cus / ntiles, 1);  // one workgroup per CU (at most `cus`)
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // a workgroup should walk several bricks, or the double buffering has nothing to overlap
    if (nbricks < 4 * nsplit) return TDX_ESHAPE;
    const size_t lds = (size_t)2 * SR_SET;
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3_wgrad_split_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3_wgrad_split_ring_kernel, dim3((unsigned)(ntiles * nsplit)), dim3(768), lds, st, (const float*)x1, C1,
                       (const float*)x2, C2, (const float*)dy, out, dbias, g, Cout, nsplit, n_ci, slab_stride);
    return tdx_launch_status();
}
