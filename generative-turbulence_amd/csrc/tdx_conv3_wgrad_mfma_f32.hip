// fp32 MFMA weight gradient of the replicate-padded 3x3x3 convolution (gfx950), fp32 parity mode.
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]
//
// Same decomposition as the bf16 kernel (tdx_conv3_wgrad_mfma.hip): a workgroup owns a 32(ci) x 32*NT(co)
// tile of all 27 taps and walks a strided subset of 4 x 8 x 8-voxel bricks; wave w accumulates taps
// w, w+4, ... (7 taps x NT tiles x 16 registers).  The GEMM's K axis is the voxel axis and
// v_mfma_f32_32x32x2_f32 takes k = 2 per issue: lanes 0-31 carry voxel (x, y, 2j), lanes 32-63 voxel
// (x, y, 2j + 1), each lane its own channel -- so fragments are plain ds_read_b32 of two adjacent
// 128-B voxel rows (all 64 banks, conflict-free), for any tap shift.  No transposed reads are needed:
// both operands are K-major in memory and the MFMA wants one (row, k) element per lane.
// A dy fragment is read once per K-step and reused by the wave's 7 taps.  An fp32 MFMA takes 64 cycles,
// so staging (next brick's global loads in flight during the MFMA phase) and LDS latency (fragments of
// step s+1 read during step s) are an order of magnitude cheaper, relatively, than in the bf16 kernel.
// Products and sums are IEEE fp32.  Partial tiles: f32 atomics into dwp[27][Cin][Cout], or plain stores
// into per-split slabs when there are few splits.
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

// brick BX x 8 x 8 with BX = 4 for NT = 1 and BX = 2 for NT = 2 (224 accumulator registers: the
// smaller brick keeps the prefetch registers of the next brick inside the 512-register budget)
#define WF_BX(NT) ((NT) == 2 ? 2 : 4)
#define WF_BY 8
#define WF_BZ 8
#define WF_HY 10
#define WF_HZ 10
#define WF_ROW 128                                 // bytes of a 32-channel fp32 voxel row
#define WF_TAPS_PER_WAVE 7

struct WgradViewF {
    int B;
    int E[3];   // extents in the kernel's local axes
    int s[3];   // voxel strides
    int ws[3];  // weight-tap strides
    int nb[3];  // bricks per axis
    int batch;  // voxels per sample
};

bool conv3_wgrad_mfma_f32_supported(int C1, int C2, int Cout) {
    const bool c1_ok = (C1 % 32) == 0 || (C2 == 0 && (C1 % 8) == 0);
    return C1 > 0 && c1_ok && (C2 % 32) == 0 && (Cout % 32) == 0;
}

template <int NT>
__global__ void __launch_bounds__(256, 1)
conv3_wgrad_mfma_f32_kernel(const float* __restrict__ x1, int C1, const float* __restrict__ x2, int C2,
                            const float* __restrict__ dy, float* __restrict__ dwp, float* __restrict__ dbias,
                            WgradViewF gv, int Cout, int nsplit, int n_ci_tiles, int64_t slab_stride) {
    constexpr int BX = WF_BX(NT);
    constexpr int WF_NVOX = BX * WF_BY * WF_BZ;          // voxels per brick
    constexpr int WF_NSTEPS = WF_NVOX / 2;               // K-steps of 2 voxels
    constexpr int WF_NHALO = (BX + 2) * WF_HY * WF_HZ;
    constexpr int WF_XBYTES = WF_NHALO * WF_ROW;
    constexpr int WF_GPLANE = WF_NVOX * WF_ROW;          // one 32-channel dy plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sG = smem + WF_XBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32;
    const int co0 = (tile / n_ci_tiles) * (32 * NT);
    const float* xs;
    int Cs, cbase;
    if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }

    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    f32x16 acc[WF_TAPS_PER_WAVE][NT];
#pragma unroll
    for (int t = 0; t < WF_TAPS_PER_WAVE; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][nt][i] = 0.f;
    const bool do_bias = dbias != nullptr && ci0 == 0;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};  // a thread always stages the same 4 dy channels (piece tid % (8 NT))

    // fragment base of this lane at step 0: halo voxel (1, 1, 1 + hh) + tap, channel r
    const unsigned char* a_base[WF_TAPS_PER_WAVE];
#pragma unroll
    for (int t = 0; t < WF_TAPS_PER_WAVE; ++t) {
        const int tap = min(wave + 4 * t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        const int toff = (ex * WF_HY + ey) * WF_HZ + ez;
        a_base[t] = sX + ((WF_HY + 1) * WF_HZ + 1 + hh + toff) * WF_ROW + r * 4;
    }
    const unsigned char* b_base = sG + hh * WF_ROW + r * 4;

    constexpr int XP = (WF_NHALO * 8 + 255) / 256;  // 16-B pieces per thread
    constexpr int GP = (WF_NVOX * 8 * NT) / 256;
    float4 xreg[XP], greg[GP];

    auto load_brick = [&](int brick) {
        int bb = brick;
        const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
        const int by = bb % gv.nb[1]; bb /= gv.nb[1];
        const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
        const int b = bb;
        const int ox0 = bx * BX, oy0 = by * WF_BY, oz0 = bz * WF_BZ;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            xreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pc < WF_NHALO * 8 && cbase + (pc & 7) * 4 < Cs) {
                const int hv = pc >> 3, q4 = pc & 7;
                const int hx = hv / (WF_HY * WF_HZ), rem = hv - hx * (WF_HY * WF_HZ);
                const int hy = rem / WF_HZ, hz = rem - hy * WF_HZ;
                const int sx = min(max(ox0 + hx - 1, 0), gv.E[0] - 1), sy = min(max(oy0 + hy - 1, 0), gv.E[1] - 1),
                          sz = min(max(oz0 + hz - 1, 0), gv.E[2] - 1);
                const int64_t vox = (int64_t)b * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
                xreg[i] = *reinterpret_cast<const float4*>(xs + vox * Cs + cbase + q4 * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            const int v = pc / (8 * NT), q4 = pc - v * (8 * NT);
            const int vx = ox0 + (v >> 6), vy = oy0 + ((v >> 3) & 7), vz = oz0 + (v & 7);
            greg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2]) {
                const int64_t vox = (int64_t)b * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
                greg[i] = *reinterpret_cast<const float4*>(dy + vox * Cout + co0 + q4 * 4);
            }
        }
    };
    auto store_brick = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            if (pc < WF_NHALO * 8) *reinterpret_cast<float4*>(sX + pc * 16) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            const int v = pc / (8 * NT), q4 = pc - v * (8 * NT);
            *reinterpret_cast<float4*>(sG + (q4 >> 3) * WF_GPLANE + v * WF_ROW + (q4 & 7) * 16) = greg[i];
            if (do_bias) { bs[0] += greg[i].x; bs[1] += greg[i].y; bs[2] += greg[i].z; bs[3] += greg[i].w; }
        }
    };

    // K-step s: voxels (x = s >> 5, y = (s >> 2) & 7, z = 2 (s & 3) + hh)
    auto step_off = [&](int s) { return (((s >> 5) * WF_HY + ((s >> 2) & 7)) * WF_HZ + 2 * (s & 3)) * WF_ROW; };
    auto read_a = [&](int soff, int t) { return *reinterpret_cast<const float*>(a_base[t] + soff); };
    auto read_b = [&](int s, float (&bf)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const float*>(b_base + nt * WF_GPLANE + 2 * s * WF_ROW);
    };

    int brick = split;
    if (brick < nbricks) load_brick(brick);
    for (; brick < nbricks; brick += nsplit) {
        __syncthreads();  // previous brick's fragment reads are done
        store_brick();
        __syncthreads();
        if (brick + nsplit < nbricks) load_brick(brick + nsplit);  // in flight during the MFMA phase

        // two register sets: while step s issues its 7 NT MFMAs from one, the fragments of step s+1
        // are read into the other
        float A0[WF_TAPS_PER_WAVE], A1[WF_TAPS_PER_WAVE], B0[NT], B1[NT];
#pragma unroll
        for (int t = 0; t < WF_TAPS_PER_WAVE; ++t) A0[t] = read_a(0, t);
        read_b(0, B0);
#pragma unroll 1
        for (int s2 = 0; s2 < WF_NSTEPS / 2; ++s2) {
            const int so = 2 * s2 + 1, sn = min(2 * s2 + 2, WF_NSTEPS - 1);
            const int off_o = step_off(so), off_n = step_off(sn);
#pragma unroll
            for (int t = 0; t < WF_TAPS_PER_WAVE; ++t) {
                A1[t] = read_a(off_o, t);
                if (t == 0) read_b(so, B1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[t], B0[nt], acc[t][nt], 0, 0, 0);
                if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
#pragma unroll
            for (int t = 0; t < WF_TAPS_PER_WAVE; ++t) {
                A0[t] = read_a(off_n, t);
                if (t == 0) read_b(sn, B0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[t], B1[nt], acc[t][nt], 0, 0, 0);
                if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
        }
    }

    // ---- merge: D[row = ci][col = co]; lane holds col r, rows (i & 3) + 8 (i >> 2) + 4 hh
#pragma unroll
    for (int t = 0; t < WF_TAPS_PER_WAVE; ++t) {
        const int ltap = wave + 4 * t;  // tap in local axes -> tap of the weight tensor
        if (ltap < 27) {
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (ci >= Cin) continue;  // partly filled last tile
                    float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + nt * 32 + r];
                    if (slab_stride) dst[(int64_t)split * slab_stride] = acc[t][nt][i];
                    else atomicAdd(dst, acc[t][nt][i]);
                }
        }
    }
    if (do_bias) {
        // threads with equal tid % (8 NT) hold partial sums of the same 4 channels
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // [256][4]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = bs[e];
        __syncthreads();
        if (tid < 32 * NT) {
            const int q4 = tid >> 2, e = tid & 3;
            float t = 0.f;
            for (int k = q4; k < 256; k += 8 * NT) t += red[k * 4 + e];
            atomicAdd(&dbias[co0 + tid], t);
        }
    }
}

int conv3_wgrad_mfma_f32_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp,
                                float* dbias, int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs,
                                int max_slabs, int* nslab_out) {
    const int Cin = C1 + C2;
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    // local axes: brick 4 x 8 x 8; the short axis goes where it leaves the fewest bricks
    const int E[3] = {X, Y, Z}, gs[3] = {Y * Z, Z, 1}, gw[3] = {9, 3, 1};
    const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
    int best = 0;
    int64_t best_n = -1;
    for (int c = 0; c < 3; ++c) {
        const int64_t n = (int64_t)ceil_div(E[cand[c][0]], WF_BX(NT)) * ceil_div(E[cand[c][1]], WF_BY) * ceil_div(E[cand[c][2]], WF_BZ);
        if (best_n < 0 || n < best_n) { best_n = n; best = c; }
    }
    WgradViewF g;
    g.B = B; g.batch = X * Y * Z;
    const int bdim[3] = {WF_BX(NT), WF_BY, WF_BZ};
    for (int k = 0; k < 3; ++k) {
        const int a = cand[best][k];
        g.E[k] = E[a]; g.s[k] = gs[a]; g.ws[k] = gw[a]; g.nb[k] = ceil_div(E[a], bdim[k]);
    }
    const int nbricks = B * g.nb[0] * g.nb[1] * g.nb[2];
    const int n_ci = (Cin + 31) / 32, n_co = Cout / (32 * NT);
    const int ntiles = n_ci * n_co;
    // one workgroup per CU (LDS): one resident wave of workgroups, at most one split per brick
    int nsplit = (256 + ntiles - 1) / ntiles;
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    const size_t lds = (size_t)(WF_BX(NT) + 2) * WF_HY * WF_HZ * WF_ROW + (size_t)NT * WF_BX(NT) * WF_BY * WF_BZ * WF_ROW;
    dim3 grid((unsigned)(ntiles * nsplit));
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
#define WF_LAUNCH(NTV)                                                                                               \
    do {                                                                                                             \
        auto kern = conv3_wgrad_mfma_f32_kernel<NTV>;                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const float*)x1, C1, (const float*)x2, C2,               \
                           (const float*)dy, out, dbias, g, Cout, nsplit, n_ci, slab_stride);                        \
    } while (0)
    if (NT == 2) WF_LAUNCH(2); else WF_LAUNCH(1);
#undef WF_LAUNCH
    return tdx_launch_status();
}
