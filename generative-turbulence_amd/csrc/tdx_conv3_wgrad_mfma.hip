// bf16 MFMA weight gradient of the replicate-padded 3x3x3 convolution (gfx950).
//
//   dW[tap][ci][co] = sum_v x[clamp(v + tap)][ci] * dy[v][co]
//
// A "TN" GEMM per tap: M = ci, N = co, K = voxels -- both operands are stored voxel-major
// (NDHWC), i.e. K-major, so every MFMA fragment is a transposed LDS read
// (ds_read_b64_tr_b16: 4 voxel rows x 16 channels per 16-lane group, delivered
// channel-major).  Rows are 64 B (32 channels), so the four consecutive-z rows a lane
// group reads are 256 contiguous bytes: conflict-free by construction.
//
// One workgroup (4 waves, one per SIMD, whole 512-register file) owns a 32(ci) x 32*NT(co)
// tile of all 27 taps and walks a strided subset of 4x8x8-voxel bricks.  Per brick it
// stages the halo'd x brick (600 voxels x 32 ch = 38 KB) and the dy brick (256 voxels x
// 32*NT ch) into LDS; wave w accumulates taps w, w+4, ... (7 taps x NT tiles x 16 regs)
// over the brick's 16 K-steps of 16 voxels, so a dy fragment is read once per K-step and
// reused by the wave's 7 taps.  Global loads for the next brick are issued before the MFMA
// phase of the current one and written to LDS after it.  Partial tiles are merged with f32
// atomics (128-B contiguous per half-wave) into dwp[27][Cin][Cout].  The bias gradient is summed
// from the dy staging registers (a thread always stages the same 8 channels).  The brick's short
// axis is put on the grid axis that leaves the fewest bricks (WgradView: local axes).
#include "tdx_common.h"
#include "tdx_conv3.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define W3_BX 4
#define W3_BY 8
#define W3_BZ 8
#define W3_HY 10
#define W3_HZ 10
#define W3_NVOX (W3_BX * W3_BY * W3_BZ)              // 256 voxels per brick
#define W3_NSTEPS (W3_NVOX / 16)                     // K-steps of 16 voxels
#define W3_NHALO ((W3_BX + 2) * W3_HY * W3_HZ)        // 1000
#define W3_XBYTES (W3_NHALO * 64)   // 64000
#define W3_GPLANE (W3_NVOX * 64)    // one 32-channel dy plane
#define W3_TAPS_PER_WAVE 7

// grid in the kernel's local axes (local axis k = global axis perm[k]; the brick is 4 x 8 x 8 in local
// axes, and the short axis is put where it leaves the fewest bricks)
struct WgradView {
    int B;
    int E[3];     // extents
    int s[3];     // voxel strides
    int ws[3];    // weight-tap strides: global tap = sum_k (e_k + 1) * ws[k]
    int nb[3];    // bricks per axis
    int batch;    // voxels per sample
};

// a single input may end in a half-filled 32-channel tile (C1 % 16 == 0): its missing channels are
// staged as zeros and their rows of dW are not written
bool conv3_wgrad_mfma_supported(int C1, int C2, int Cout) {
    const bool c1_ok = (C1 % 32) == 0 || (C2 == 0 && (C1 % 16) == 0);
    return C1 > 0 && c1_ok && (C2 % 32) == 0 && (Cout % 32) == 0;
}

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base_lo, const unsigned char* base_hi) {
    // two transposed 4-row reads -> 8 consecutive k for this lane's column
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base_hi));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, r);
}

// Diagnostic builds only (tools/micro/wgrad_stamp.hip defines W3_STAMPS): s_memtime stamps of brick iterations 4 .. 11
// of every wave per phase, kept in spare LDS and dumped to w3_stamps_dev at the end; the product build carries none of it.
#ifdef W3_STAMPS
#define W3_NSTAMP 64
__device__ unsigned long long* w3_stamps_dev;
#define W3_T()                                                                              \
    do {                                                                                    \
        if (it_ >= 4 && it_ < 12) {                                                         \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
            if (lane == 0 && nst_ < W3_NSTAMP) sStamp_[wave * W3_NSTAMP + nst_] = t_;       \
            ++nst_;                                                                         \
        }                                                                                   \
    } while (0)
#else
#define W3_T() do { } while (0)
#endif

// HF: operand format (H16<HF>: bf16 or fp16 words behind the bf16-typed pointers)
template <int NT, bool HF>
__global__ void __launch_bounds__(256, NT == 2 ? 1 : 2)
conv3_wgrad_mfma_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2,
                        const bf16* __restrict__ dy, float* __restrict__ dwp, float* __restrict__ dbias, WgradView gv,
                        int Cout, int nsplit, int n_ci_tiles, int64_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sG = smem + W3_XBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = C1 + C2;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32;
    const int co0 = (tile / n_ci_tiles) * (32 * NT);
    const bf16* xs;
    int Cs, cbase;
    if (ci0 < C1) { xs = x1; Cs = C1; cbase = ci0; } else { xs = x2; Cs = C2; cbase = ci0 - C1; }

    const int nbricks = gv.B * gv.nb[0] * gv.nb[1] * gv.nb[2];

    // ---- fragment lane geometry (see file header)
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;   // byte offset of this lane's 4 columns in a 64-B row
    const int kh = g >> 1;                            // which 8-voxel half of the 16-voxel K-step

    f32x16 acc[W3_TAPS_PER_WAVE][NT];
#pragma unroll
    for (int t = 0; t < W3_TAPS_PER_WAVE; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][nt][i] = 0.f;
    const bool do_bias = dbias != nullptr && ci0 == 0;
    float bs[8];  // bias gradient: this thread always stages the same 8 dy channels (chunk tid % (4 NT))
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    // halo offsets (in voxels) of this wave's taps
    int toff[W3_TAPS_PER_WAVE];
#pragma unroll
    for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) {
        const int tap = min(wave + 4 * t, 26);
        const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
        toff[t] = (ex * W3_HY + ey) * W3_HZ + ez;
    }

    // per-tap fragment base address of this lane at step 0: halo voxel (1, kh + 1, q + 1) + tap
    const unsigned char* a_base[W3_TAPS_PER_WAVE];
#pragma unroll
    for (int t = 0; t < W3_TAPS_PER_WAVE; ++t)
        a_base[t] = sX + ((W3_HY + kh + 1) * W3_HZ + (q + 1) + toff[t]) * 64 + col_off;

    constexpr int XP = (W3_NHALO * 4 + 255) / 256;  // 16 pieces of 16 B per thread
    constexpr int GP = (W3_NVOX * 4 * NT) / 256;     // 8*NT pieces per thread
    uint4 xreg[XP], greg[GP];

    auto load_brick = [&](int brick) {
        int bb = brick;
        const int bz = bb % gv.nb[2]; bb /= gv.nb[2];
        const int by = bb % gv.nb[1]; bb /= gv.nb[1];
        const int bx = bb % gv.nb[0]; bb /= gv.nb[0];
        const int b = bb;
        const int ox0 = bx * W3_BX, oy0 = by * W3_BY, oz0 = bz * W3_BZ;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            xreg[i] = make_uint4(0, 0, 0, 0);
            if (pc < W3_NHALO * 4 && cbase + (pc & 3) * 8 < Cs) {
                const int hv = pc >> 2, q4 = pc & 3;
                const int hx = hv / (W3_HY * W3_HZ), rem = hv - hx * (W3_HY * W3_HZ);
                const int hy = rem / W3_HZ, hz = rem - hy * W3_HZ;
                const int sx = min(max(ox0 + hx - 1, 0), gv.E[0] - 1), sy = min(max(oy0 + hy - 1, 0), gv.E[1] - 1),
                          sz = min(max(oz0 + hz - 1, 0), gv.E[2] - 1);
                const int64_t vox = (int64_t)b * gv.batch + sx * gv.s[0] + sy * gv.s[1] + sz * gv.s[2];
                xreg[i] = *reinterpret_cast<const uint4*>(xs + vox * Cs + cbase + q4 * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            const int v = pc / (4 * NT), q8 = pc - v * (4 * NT);
            const int vx = ox0 + (v >> 6), vy = oy0 + ((v >> 3) & 7), vz = oz0 + (v & 7);
            greg[i] = make_uint4(0, 0, 0, 0);
            if (vx < gv.E[0] && vy < gv.E[1] && vz < gv.E[2]) {
                const int64_t vox = (int64_t)b * gv.batch + vx * gv.s[0] + vy * gv.s[1] + vz * gv.s[2];
                greg[i] = *reinterpret_cast<const uint4*>(dy + vox * Cout + co0 + q8 * 8);
            }
        }
    };
    auto store_brick = [&]() {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int pc = tid + i * 256;
            if (pc < W3_NHALO * 4) *reinterpret_cast<uint4*>(sX + pc * 16) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < GP; ++i) {
            const int pc = tid + i * 256;
            const int v = pc / (4 * NT), q8 = pc - v * (4 * NT);
            *reinterpret_cast<uint4*>(sG + (q8 >> 2) * W3_GPLANE + v * 64 + (q8 & 3) * 16) = greg[i];
            if (do_bias) {
                const unsigned wds[4] = {greg[i].x, greg[i].y, greg[i].z, greg[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bs[2 * e] += H16<HF>::lo(wds[e]);
                    bs[2 * e + 1] += H16<HF>::hi(wds[e]);
                }
            }
        }
    };

#ifdef W3_STAMPS
    unsigned long long* sStamp_ = reinterpret_cast<unsigned long long*>(smem + W3_XBYTES + NT * W3_GPLANE);
    int nst_ = 0, it_ = 0;
#endif
    int brick = split;
    if (brick < nbricks) load_brick(brick);
    for (; brick < nbricks; brick += nsplit) {
        W3_T();  // iteration top
        __syncthreads();  // previous brick's fragment reads are done
        W3_T();  // barrier 1 passed (includes the wait for the staged loads)
        store_brick();
#ifdef W3_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        W3_T();  // LDS stores done
        __syncthreads();
        W3_T();  // barrier 2 passed
        if (brick + nsplit < nbricks) load_brick(brick + nsplit);  // in flight during the MFMA phase
        W3_T();  // next brick's loads issued

        // K-step s: voxels (x = s >> 2, y = 2 (s & 3) + kh, z = q (+4)).  One wave per SIMD, so the
        // LDS latency has to be hidden inside the wave: while step s issues its 7 x NT MFMAs from
        // one register set, the 7 x-fragments and the dy fragments of step s+1 are read into the
        // other set (a whole step, ~450 cycles, of cover).  Two steps per loop trip keep every
        // register index static.
        auto read_b = [&](int s, bf16x8 (&bf)[NT]) {
            const unsigned char* bp = sG + (16 * s + 8 * kh + q) * 64 + col_off;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = tr_frag(bp + nt * W3_GPLANE, bp + nt * W3_GPLANE + 4 * 64);
        };
        // byte offset of step s inside the halo brick (x = s >> 2, y = 2 (s & 3))
        auto step_off = [&](int s) { return ((s >> 2) * W3_HY + 2 * (s & 3)) * W3_HZ * 64; };
        auto read_a = [&](int soff, int t) {
            const unsigned char* ap = a_base[t] + soff;
            return tr_frag(ap, ap + 4 * 64);
        };
        if constexpr (NT == 2) {
            bf16x8 A0[W3_TAPS_PER_WAVE], A1[W3_TAPS_PER_WAVE], B0[NT], B1[NT];
    #pragma unroll
            for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) A0[t] = read_a(0, t);
            read_b(0, B0);
    #pragma unroll 1
            for (int s2 = 0; s2 < W3_NSTEPS / 2; ++s2) {
                const int so = 2 * s2 + 1, sn = min(2 * s2 + 2, W3_NSTEPS - 1);
                const int off_o = step_off(so), off_n = step_off(sn);
    #pragma unroll
                for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) {  // even step: compute set 0, fetch set 1
                    A1[t] = read_a(off_o, t);
                    if (t == 0) read_b(so, B1);
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = H16<HF>::mfma(A0[t], B0[nt], acc[t][nt]);
                    if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * NT, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                }
    #pragma unroll
                for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) {  // odd step: compute set 1, fetch set 0
                    A0[t] = read_a(off_n, t);
                    if (t == 0) read_b(sn, B0);
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = H16<HF>::mfma(A1[t], B1[nt], acc[t][nt]);
                    if (t == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * NT, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                }
            }
        } else {
            // NT = 1: 112 accumulator registers -> two workgroups per CU cover each other's LDS
            // latency and staging; keep the register count under 256 with a plain loop.
#pragma unroll 2
            for (int s = 0; s < W3_NSTEPS; ++s) {
                bf16x8 bf[NT];
                read_b(s, bf);
                const int soff = step_off(s);
#pragma unroll
                for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) {
                    const bf16x8 af = read_a(soff, t);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[t][nt] = H16<HF>::mfma(af, bf[nt], acc[t][nt]);
                }
            }
        }
        W3_T();  // MFMAs of the brick issued
#ifdef W3_STAMPS
        ++it_;
#endif
    }
#ifdef W3_STAMPS
    if (lane == 0 && w3_stamps_dev != nullptr) {
        unsigned long long* rec = w3_stamps_dev + ((size_t)blockIdx.x * 4 + wave) * (W3_NSTAMP + 1);
        rec[0] = nst_;
        for (int i = 0; i < W3_NSTAMP; ++i) rec[1 + i] = sStamp_[wave * W3_NSTAMP + i];
    }
#endif

    // ---- merge: D[row = ci][col = co]; lane holds col (lane & 31), rows (i&3) + 8 (i>>2) + 4 (lane>>5)
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < W3_TAPS_PER_WAVE; ++t) {
        const int ltap = wave + 4 * t;  // tap in local axes -> tap of the weight tensor
        if (ltap < 27) {
            const int tap = (ltap / 9) * gv.ws[0] + ((ltap / 3) % 3) * gv.ws[1] + (ltap % 3) * gv.ws[2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ci = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (ci >= Cin) continue;  // half-filled last tile
                    float* dst = &dwp[((int64_t)tap * Cin + ci) * Cout + co0 + nt * 32 + r];
                    // few splits (deep layers): every split owns a slab and stores plainly, the unpack
                    // kernel adds the slabs; many splits (fine levels): f32 atomics into one table
                    if (slab_stride) dst[(int64_t)split * slab_stride] = acc[t][nt][i];
                    else atomicAdd(dst, acc[t][nt][i]);
                }
        }
    }
    if (do_bias) {
        // threads with equal tid % (4 NT) hold partial sums of the same 8 channels
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // [256][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 32 * NT) {
            const int q8 = tid >> 3, e = tid & 7;
            float t = 0.f;
            for (int k = q8; k < 256; k += 4 * NT) t += red[k * 8 + e];
            atomicAdd(&dbias[co0 + tid], t);
        }
    }
}

int conv3_wgrad_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias,
                            int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs,
                            int* nslab_out, bool hf) {
    {   // deep U-Net levels: packed-K kernel (tdx_conv3_wgrad_small.hip); TDX_ESHAPE = not such a case
        int rs = conv3_wgrad_small_launch(x1, C1, x2, C2, dy, dwp, dbias, B, X, Y, Z, Cout, st, slabs, max_slabs, nslab_out, hf);
        if (rs != TDX_ESHAPE) return rs;
#ifndef W3_STAMPS
        // fine levels: the producer / consumer form (8 computing + 4 loader waves)
        rs = conv3_wgrad_ring_launch(x1, C1, x2, C2, dy, dwp, dbias, B, X, Y, Z, Cout, st, slabs, max_slabs, nslab_out, hf);
        if (rs != TDX_ESHAPE) return rs;
#endif
    }
    const int Cin = C1 + C2;
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    // local axes: brick 4 x 8 x 8; the short axis goes where it leaves the fewest bricks
    const int E[3] = {X, Y, Z}, gs[3] = {Y * Z, Z, 1}, gw[3] = {9, 3, 1};
    const int cand[3][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}};
    static const bool no_perm = getenv("TDX_CONV3_PERM") && atoi(getenv("TDX_CONV3_PERM")) == 0;  // A/B switch
    int best = 0;
    int64_t best_n = -1;
    for (int c = 0; c < (no_perm ? 1 : 3); ++c) {
        const int64_t n = (int64_t)ceil_div(E[cand[c][0]], W3_BX) * ceil_div(E[cand[c][1]], W3_BY) * ceil_div(E[cand[c][2]], W3_BZ);
        if (best_n < 0 || n < best_n) { best_n = n; best = c; }
    }
    WgradView g;
    g.B = B; g.batch = X * Y * Z;
    const int bdim[3] = {W3_BX, W3_BY, W3_BZ};
    for (int k = 0; k < 3; ++k) {
        const int a = cand[best][k];
        g.E[k] = E[a]; g.s[k] = gs[a]; g.ws[k] = gw[a]; g.nb[k] = ceil_div(E[a], bdim[k]);
    }
    const int nbricks = B * g.nb[0] * g.nb[1] * g.nb[2];
    const int n_ci = (Cin + 31) / 32, n_co = Cout / (32 * NT);
    const int ntiles = n_ci * n_co;
    // one workgroup per CU (224 accumulator registers -> one wave per SIMD): aim at 256
    // workgroups overall, at most one split per brick.  Fewer, longer workgroups also mean fewer
    // bytes of f32 atomics at the end (221 KB per workgroup).
    int nsplit = ((NT == 2 ? 256 : 512) + ntiles - 1) / ntiles;
    if (nsplit > nbricks) nsplit = nbricks;
    if (nsplit < 1) nsplit = 1;
    // TDX_DETERMINISTIC: never the atomic merge -- hold the K splits to the slabs the workspace has (added in order by the unpack kernel)
    if (tdx_deterministic() && slabs != nullptr && nsplit > max_slabs) nsplit = max_slabs > 0 ? max_slabs : 1;
    size_t lds = W3_XBYTES + (size_t)NT * W3_GPLANE;
#ifdef W3_STAMPS
    lds += 4 * W3_NSTAMP * 8;
#endif
    dim3 grid((unsigned)(ntiles * nsplit));
    // slab mode: every (tile, split) pair stores its whole partial tile, so the slabs need no zeroing
    const bool use_slabs = slabs != nullptr && nsplit <= max_slabs;
    const int64_t slab_stride = use_slabs ? (int64_t)27 * Cin * Cout : 0;
    float* out = use_slabs ? slabs : dwp;
    if (nslab_out) *nslab_out = use_slabs ? nsplit : 0;
#define W3_LAUNCH(NTV, HFV)                                                                                          \
    do {                                                                                                             \
        auto kern = conv3_wgrad_mfma_kernel<NTV, HFV>;                                                                  \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const bf16*)x1, C1, (const bf16*)x2, C2, (const bf16*)dy, \
                           out, dbias, g, Cout, nsplit, n_ci, slab_stride);                               \
    } while (0)
    if (NT == 2) { if (hf) W3_LAUNCH(2, true); else W3_LAUNCH(2, false); }
    else { if (hf) W3_LAUNCH(1, true); else W3_LAUNCH(1, false); }
#undef W3_LAUNCH
    return tdx_launch_status();
}
