// Diagnostic build of the bf16 brick conv kernel's main loop (tdx_conv3_mfma.hip, NT = 2, main bricks, replicate
// padding) with s_memtime stamps at the phase boundaries of every workgroup.  NOT product code: it exists to show
// where a workgroup's cycles go (wait for staged loads / LDS stores / barriers / MFMA phase / epilogue) and how the
// two co-resident workgroups of a CU interleave.  Build: hipcc -O3 --offload-arch=gfx950 conv3_stamp.hip -o conv3_stamp
// Run: ./conv3_stamp [Cin] [NT] > stamps.txt
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef __hip_bfloat16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define NSTAMP 48
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int xcd = bid & 7, k = bid >> 3;
    const int base = n >> 3, rem = n & 7;
    return xcd * base + min(xcd, rem) + k;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const unsigned*>(&h);
}

template <int NT>
__global__ void __launch_bounds__(256, 2)
conv_stamp(const bf16* __restrict__ x1, int Cin, const bf16* __restrict__ wp, bf16* __restrict__ y, int X, int Y, int Z,
           int Cout, unsigned long long* __restrict__ dbg) {
    constexpr int BN = NT * 32;
    constexpr int BX = 4, BY = 8, BZ = 8, HX = 6, HY = 10, HZ = 10, SZ = 12;
    constexpr int NHALO = HX * HY * HZ;
    constexpr int APLANE = HX * HY * SZ * 16 + 64;
    constexpr int BRICK_BYTES = 2 * APLANE;
    constexpr int B_PLANE = 27 * BN * 16 + 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BRICK_BYTES;
    unsigned long long* sT = reinterpret_cast<unsigned long long*>(smem + BRICK_BYTES + 2 * B_PLANE);  // [4 waves][NSTAMP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int ns = 0;
#define STAMP()                                                    \
    do {                                                           \
        unsigned long long t_ = now();                             \
        if (lane == 0 && ns < NSTAMP) sT[wave * NSTAMP + ns] = t_; \
        ++ns;                                                      \
    } while (0)
    STAMP();  // 0: start
    int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    const int nb2 = Z / BZ, nb1 = Y / BY, nb0 = X / BX;
    const int b2 = bid % nb2; bid /= nb2;
    const int b1 = bid % nb1; bid /= nb1;
    const int b0 = bid % nb0; bid /= nb0;
    const int b = bid;
    const int n0 = blockIdx.y * BN;
    const int o0 = b0 * BX, o1 = b1 * BY, o2 = b2 * BZ;
    constexpr int A_PIECES = NHALO * 2, A_PER_THREAD = (A_PIECES + 255) / 256;
    constexpr int B_PIECES = 27 * BN * 2, B_PER_THREAD = (B_PIECES + 255) / 256;
    int a_src[A_PER_THREAD], a_dst[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int p = tid + i * 256;
        a_dst[i] = -1; a_src[i] = -1;
        if (p < A_PIECES) {
            const int hv = ((p >> 3) << 2) + (p & 3), half = (p >> 2) & 1;
            const int hx = hv / (HY * HZ), rem = hv - hx * (HY * HZ);
            const int hy = rem / HZ, hz = rem - hy * HZ;
            a_dst[i] = half * APLANE + ((hx * HY + hy) * SZ + hz) * 16;
            int s0 = min(max(o0 + hx - 1, 0), X - 1), s1 = min(max(o1 + hy - 1, 0), Y - 1), s2 = min(max(o2 + hz - 1, 0), Z - 1);
            a_src[i] = ((s0 * Y + s1) * Z + s2) * 2 + half;
        }
    }
    const int64_t batch_vox = (int64_t)b * X * Y * Z;
    const int b_half = (tid >> 2) & 1;
    const int b_row0 = ((tid >> 3) << 2) + (tid & 3);
    const int b_goff = ((b_row0 / BN) * Cout + (b_row0 % BN)) * 16 + b_half * 8;
    const int b_dst = b_half * B_PLANE + b_row0 * 16;
    uint4 areg[A_PER_THREAD], breg[B_PER_THREAD];
    auto load_slice = [&](int c) {
        const bf16* xs = x1 + batch_vox * Cin + c * 16;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            areg[i] = make_uint4(0, 0, 0, 0);
            if (a_src[i] >= 0) areg[i] = *reinterpret_cast<const uint4*>(xs + (int64_t)(a_src[i] >> 1) * Cin + (a_src[i] & 1) * 8);
        }
        const bf16* wc = wp + (int64_t)c * 27 * Cout * 16 + (int64_t)n0 * 16 + b_goff;
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i) {
            breg[i] = make_uint4(0, 0, 0, 0);
            if (b_row0 + 128 * i < 27 * BN) breg[i] = *reinterpret_cast<const uint4*>(wc + (int64_t)i * (128 / BN) * Cout * 16);
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(sA + a_dst[i]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_PER_THREAD; ++i)
            if (b_row0 + 128 * i < 27 * BN) *reinterpret_cast<uint4*>(sB + b_dst + i * 2048) = breg[i];
    };
    int a_h[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a_h[mt] = ((wave + 1) * HY + (4 * mt + (r & 3) + 1)) * SZ + ((r >> 2) + 1);
    int b_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b_off[nt] = hh * B_PLANE + (nt * 32 + r) * 16;
    f32x16 acc[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;
    const int nchunks = Cin / 16;
    load_slice(0);
    STAMP();  // 1: first loads issued
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        STAMP();  // 2 + 5c: barrier 1 passed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP();  // 3 + 5c: staged loads have landed in registers
        store_slice();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP();  // 4 + 5c: LDS stores done (this wave)
        __syncthreads();
        STAMP();  // 5 + 5c: barrier 2 passed
        if (c + 1 < nchunks) load_slice(c + 1);
        bf16x8 xf[2][2], wf[2][NT];
        auto read_frags = [&](int tap, int buf) {
            const int ex = tap / 9 - 1, ey = (tap / 3) % 3 - 1, ez = tap % 3 - 1;
            const int toff = (ex * HY + ey) * SZ + ez;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) xf[buf][mt] = *reinterpret_cast<const bf16x8*>(sA + hh * APLANE + (a_h[mt] + toff) * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[buf][nt] = *reinterpret_cast<const bf16x8*>(sB + tap * (BN * 16) + b_off[nt]);
        };
        read_frags(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NT, 0);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) read_frags(tap + 1, (tap + 1) & 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap & 1][nt], xf[tap & 1][mt], acc[nt][mt], 0, 0, 0);
            if (tap + 1 < 27) {
#pragma unroll
                for (int k = 0; k < 2 * NT; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (k < 2 + NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
            }
        }
        asm volatile("" ::: "memory");
        STAMP();  // 6 + 5c: MFMAs of the slice issued
    }
    __syncthreads();
    STAMP();  // epilogue start (all waves past their MFMAs)
    unsigned char* sO = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * j + 4 * hh;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int v = (wave * BY + 4 * mt + (r & 3)) * 8 + (r >> 2);
                const unsigned lo = pack_bf16x2(acc[nt][mt][4 * j], acc[nt][mt][4 * j + 1]);
                const unsigned hi = pack_bf16x2(acc[nt][mt][4 * j + 2], acc[nt][mt][4 * j + 3]);
                const int c = ch >> 3;
                const int addr = BN == 64 ? v * 128 + ((c ^ (v & 7)) << 4) : v * 64 + ((c ^ ((v >> 1) & 3)) << 4);
                *reinterpret_cast<uint2*>(sO + addr + (ch & 7) * 2) = make_uint2(lo, hi);
            }
        }
    __syncthreads();
    STAMP();  // tile in LDS
    constexpr int CHUNKS = BN / 8;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int p = tid + i * 256;
        const int v = p / CHUNKS, cidx = p % CHUNKS;
        const int c0 = o0 + (v >> 3) / BY, c1 = o1 + (v >> 3) % BY, c2 = o2 + (v & 7);
        const int addr = BN == 64 ? v * 128 + ((cidx ^ (v & 7)) << 4) : v * 64 + ((cidx ^ ((v >> 1) & 3)) << 4);
        uint4 val = *reinterpret_cast<const uint4*>(sO + addr);
        const int64_t ov = batch_vox + (c0 * Y + c1) * Z + c2;
        *reinterpret_cast<uint4*>(y + ov * Cout + n0 + cidx * 8) = val;
    }
    STAMP();  // stores issued
    // dump: record = [hw id, realtime, stamps...] per wave
    if (lane == 0) {
        const size_t rec = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * (NSTAMP + 2);
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[rec] = ((unsigned long long)xcc << 32) | hwid;
        dbg[rec + 1] = ns;
        for (int i = 0; i < NSTAMP; ++i) dbg[rec + 2 + i] = sT[wave * NSTAMP + i];
    }
}

__global__ void fill_rand(unsigned* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // two bf16 in [-1, 1): exponent 0x3f (126/127), random mantissa and sign
        const unsigned a = (h & 0x807f) | 0x3f00, b = ((h >> 16) & 0x807f) | 0x3f00;
        p[i] = a | (b << 16);
    }
}

template <int NT>
static void run(int Cin, int Cout, int B, int X, int Y, int Z) {
    constexpr int BN = NT * 32;
    const size_t nx = (size_t)B * X * Y * Z * Cin, nw = (size_t)27 * Cin * Cout, ny = (size_t)B * X * Y * Z * Cout;
    bf16 *x, *w, *y;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2);
    fill_rand<<<2048, 256>>>((unsigned*)x, nx / 2, 1);
    fill_rand<<<256, 256>>>((unsigned*)w, nw / 2, 2);
    const int nblk = B * (X / 4) * (Y / 8) * (Z / 8);
    const dim3 grid(nblk, Cout / BN);
    const size_t nrec = (size_t)nblk * grid.y * 4 * (NSTAMP + 2);
    unsigned long long* dbg;
    hipMalloc(&dbg, nrec * 8);
    const size_t lds = (size_t)2 * (6 * 10 * 12 * 16 + 64) + (size_t)2 * (27 * BN * 16 + 64) + 4 * NSTAMP * 8;
    auto kern = conv_stamp<NT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 40; ++it) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, Cin, w, y, X, Y, Z, Cout, dbg);  // warm the clock state
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, x, Cin, w, y, X, Y, Z, Cout, dbg);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double fl = 54.0 * Cin * Cout * B * X * Y * Z;
    printf("# Cin %d Cout %d NT %d grid %dx%dx%d B %d: %.3f ms/launch, %.0f TFLOP/s, lds %zu, %d workgroups\n", Cin, Cout, NT, X, Y,
           Z, B, ms, fl / ms / 1e9, lds, nblk * (int)grid.y);
    std::vector<unsigned long long> h(nrec);
    hipMemcpy(h.data(), dbg, nrec * 8, hipMemcpyDeviceToHost);
    // phase statistics over all waves
    const int nch = Cin / 16;
    const char* names[5] = {"barrier1->loads landed", "LDS stores", "barrier2", "load issue + MFMA phase", "->next barrier1"};
    std::vector<double> sums(8, 0.0);
    std::vector<std::vector<long long>> all(8);
    double life = 0, pro = 0, epi1 = 0, epi2 = 0, epi3 = 0;
    size_t nw_ = 0;
    for (size_t wv = 0; wv < (size_t)nblk * grid.y * 4; ++wv) {
        const unsigned long long* s = &h[wv * (NSTAMP + 2) + 2];
        const int last = 2 + 5 * nch;  // epilogue start index
        life += (double)(s[last + 2] - s[0]);
        pro += (double)(s[2] - s[0]);
        for (int c = 0; c < nch; ++c) {
            const unsigned long long* q = s + 2 + 5 * c;
            sums[0] += (double)(q[1] - q[0]); sums[1] += (double)(q[2] - q[1]); sums[2] += (double)(q[3] - q[2]);
            sums[3] += (double)(q[4] - q[3]); sums[4] += (double)(q[5] - q[4]);
            all[3].push_back((long long)(q[4] - q[3]));
            all[0].push_back((long long)(q[1] - q[0]));
        }
        epi1 += (double)(s[last + 1] - s[last]); epi2 += (double)(s[last + 2] - s[last + 1]);
        (void)epi3;
        ++nw_;
    }
    printf("# per wave averages (shader cycles): lifetime %.0f, prologue (start -> first barrier) %.0f\n", life / nw_, pro / nw_);
    for (int k = 0; k < 5; ++k) printf("#   per slice: %-28s %.0f\n", names[k], sums[k] / (nw_ * nch));
    printf("#   epilogue: accumulators -> LDS tile %.0f, tile -> global stores issued %.0f\n", epi1 / nw_, epi2 / nw_);
    for (int k : {0, 3}) {
        auto& v = all[k];
        std::sort(v.begin(), v.end());
        printf("#   %s percentiles: p10 %lld p50 %lld p90 %lld p99 %lld\n", names[k], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10],
               v[v.size() * 99 / 100]);
    }
    // raw records of the workgroups on one CU (first 24 workgroups that ran on the CU of workgroup 0), for a timeline
    const unsigned long long id0 = h[0] & 0xffffffff0fffff00ull;  // drop wave / simd bits (low 8) and queue bits
    int shown = 0;
    for (size_t g = 0; g < (size_t)nblk * grid.y && shown < 24; ++g) {
        const unsigned long long* rec = &h[g * 4 * (NSTAMP + 2)];
        if ((rec[0] & 0xffffffff0fffff00ull) != id0) continue;
        printf("wg %zu hw %llx:", g, rec[0]);
        for (int i = 0; i < 2 + 5 * nch + 3; ++i) printf(" %llu", rec[2 + i] - h[2]);
        printf("\n");
        ++shown;
    }
    hipFree(x); hipFree(w); hipFree(y); hipFree(dbg);
}

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 64, NT = argc > 2 ? atoi(argv[2]) : 2;
    if (NT == 2) run<2>(Cin, 64, 6, 192, 64, 48);
    else run<1>(Cin, 32, 6, 192, 64, 48);
    return 0;
}
