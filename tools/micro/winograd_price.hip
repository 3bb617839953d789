// Pricing, part 2 (GPU): the INNER LOOP of a Winograd-domain 3x3x3 convolution against the direct one, as LDS-fed fp16 MFMA
// loops in the mould of mfma_shape.hip (random operands re-read from LDS, 8 waves per CU, one workgroup per CU, no global
// traffic): what the matrix pipe, the LDS and the vector ALU make of the two instruction mixes -- before anything is built.
//
// Unit of work (both loops): 8 input channels x all taps for ONE wave's output tile of 64 voxels x 64 output channels --
// what conv3_ring_kernel<2, ...> does per 8-channel unit (tdx_conv3_ring.hip; reference op ddpm.py:164).
//   direct   : 14 K steps (27 taps in pairs + 1 dummy), each 2 x fragments + 2 w fragments (ds_read_b128) -> 2 x 2 MFMAs:
//              56 MFMAs, 56 fragment reads (1 KiB of LDS per MFMA)
//   F(2,3) z : rows of the M tile are z PAIRS (one 32-row tile = the same 64 voxels); per (dx, dy) tap PAIR (5 pairs: 9 taps +
//              1 dummy): 4 raw x fragments d0..d3 (z offsets -1 .. +2 of the pair), transformed BETWEEN ds_read and MFMA in
//              packed fp16 (B^T d: d0 - d2, d1 + d2, d2 - d1, d1 - d3 = 16 v_pk_add_f16), 4 x 2 transformed-weight fragments,
//              4 x 2 MFMAs into 8 accumulator tiles (2x the accumulators): 40 MFMAs, 60 fragment reads (1.5 KiB per MFMA),
//              80 packed adds per unit.  (The output transform A^T m runs once per brick, not per unit: not in the loop.)
//   variants : the same without the packed adds (what the LDS alone costs), and with the weight fragments held in registers
//              across the four transform indices' two N tiles is impossible (8 distinct fragments) -- so none.
// The transform cannot ride in the LDS-DMA staging (global_load_lds moves bytes), and an HBM-side transform would write and
// re-read 2x the activation bytes (4 transformed values per 2 inputs): 2 x 2 x 75 MB per 64-channel level-0 tensor at B = 6 =
// +0.06 ms per layer at 5 TB/s plus a pass nobody has: it loses on bytes before it starts.
// Build: hipcc -O3 --offload-arch=gfx950 winograd_price.hip -o winograd_price
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_ENT 8192  // 128 KiB of operand data

__device__ __forceinline__ f16x8 ld(const uint4* sm, int i) { return __builtin_bit_cast(f16x8, sm[i & (LDS_ENT - 1)]); }

__global__ void __launch_bounds__(512, 1) direct_loop(float* out, const uint4* src, int units) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < LDS_ENT; i += 512) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) b[i] = 0.f;
    int off = wave * 64 + lane;
    for (int u = 0; u < units; ++u) {
#pragma unroll
        for (int p = 0; p < 14; ++p) {
            off += 257;
            f16x8 A[2], B[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { A[j] = ld(sm, off + 64 * j); B[j] = ld(sm, off + 4096 + 64 * j); }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B[n], A[m], acc[m][n], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}

template <bool TRANSFORM>
__global__ void __launch_bounds__(512, 1) wino_loop(float* out, const uint4* src, int units) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < LDS_ENT; i += 512) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4][2];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) b[i] = 0.f;
    int off = wave * 64 + lane;
    for (int u = 0; u < units; ++u) {
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            off += 257;
            f16x8 d[4], V[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = ld(sm, off + 64 * j);
            if (TRANSFORM) { V[0] = d[0] - d[2]; V[1] = d[1] + d[2]; V[2] = d[2] - d[1]; V[3] = d[1] - d[3]; }
            else { V[0] = d[0]; V[1] = d[1]; V[2] = d[2]; V[3] = d[3]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const f16x8 W = ld(sm, off + 4096 + 64 * (2 * i + n));
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, V[i], acc[i][n], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}

// The same loop, software-pipelined by hand: the raw fragments of group g + 1 are read and transformed WHILE the 8 MFMAs of
// group g issue (two packed adds and one or two fragment reads pinned behind every MFMA with sched_group_barrier), so that
// the vector ALU works in the matrix pipe's shadow instead of in front of it.
__global__ void __launch_bounds__(512, 1) wino_loop_pipelined(float* out, const uint4* src, int units) {
    extern __shared__ uint4 sm[];
    for (int i = threadIdx.x; i < LDS_ENT; i += 512) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4][2];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) b[i] = 0.f;
    int off = wave * 64 + lane;
    f16x8 d[4], V[4], Vn[4], W[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = ld(sm, off + 64 * j);
    V[0] = d[0] - d[2]; V[1] = d[1] + d[2]; V[2] = d[2] - d[1]; V[3] = d[1] - d[3];
    W[0][0] = ld(sm, off + 4096); W[0][1] = ld(sm, off + 4096 + 64);
    const int groups = units * 5;
    for (int g = 0; g < groups; ++g) {
        const int noff = off + 257;
        // raw fragments of the next group
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = ld(sm, noff + 64 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // weight fragments of index i + 1 (or of the next group's index 0) while index i multiplies
            const int wi = i < 3 ? off + 4096 + 64 * (2 * (i + 1)) : noff + 4096;
            W[(i + 1) & 1][0] = ld(sm, wi); W[(i + 1) & 1][1] = ld(sm, wi + 64);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i & 1][0], V[i], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i & 1][1], V[i], acc[i][1], 0, 0, 0);
            // one transformed fragment of the next group per index (4 packed adds) in the shadow of these two MFMAs
            if (i == 0) Vn[0] = d[0] - d[2];
            if (i == 1) Vn[1] = d[1] + d[2];
            if (i == 2) Vn[2] = d[2] - d[1];
            if (i == 3) Vn[3] = d[1] - d[3];
            if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // ds reads (the next group's raw fragments ride with index 0)
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) V[i] = Vn[i];
        off = noff;
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
static double run(const char* name, K kern, const uint4* src, int units, int mfma_per_unit) {
    float* out; (void)hipMalloc(&out, 4);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ENT * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS_ENT * 16, 0, out, src, units);
    (void)hipEventRecord(e0);
    const int reps = 40;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS_ENT * 16, 0, out, src, units);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_unit_ns = ms * 1e6 / reps / units;                       // per wave-unit (8 waves per CU run concurrently)
    const double issued = 2.0 * 32 * 32 * 16 * mfma_per_unit * (double)units * 8 * 256 * reps / ms / 1e9;
    const double algorithmic = 2.0 * 64 * 64 * 8 * 27 * (double)units * 8 * 256 * reps / ms / 1e9;  // 64 voxels x 64 couts x 8 cins x 27 taps
    printf("%-54s %7.3f ms  %7.1f ns per unit  issued %7.1f TFLOP/s  algorithmic %7.1f TFLOP/s\n", name, ms / reps, per_unit_ns, issued, algorithmic);
    (void)hipFree(out);
    return per_unit_ns;
}

int main() {
    const size_t n = (size_t)LDS_ENT * 16;
    unsigned short* h = (unsigned short*)malloc(n);
    srand(1);
    // fp16 values of magnitude 0.25 .. 2 with random signs and mantissas (finite sums over any K)
    for (size_t i = 0; i < n / 2; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((13 + (rand() % 3)) << 10) | (rand() & 1023));
    uint4* src; (void)hipMalloc(&src, n); (void)hipMemcpy(src, h, n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        const double d = run("direct: 14 steps x (2 + 2 reads, 2 x 2 MFMAs)", direct_loop, src, 600, 56);
        const double w = run("F(2,3) z: 5 x (4 + 8 reads, 16 pk adds, 4 x 2 MFMAs)", wino_loop<true>, src, 600, 40);
        const double l = run("F(2,3) z without the packed adds (LDS mix only)", wino_loop<false>, src, 600, 40);
        const double pp = run("F(2,3) z, software-pipelined (adds behind the MFMAs)", wino_loop_pipelined, src, 600, 40);
        printf("  -> same outputs per unit: Winograd-z inner loop %.3fx the direct one (pipelined by hand %.3fx; %.3fx without its adds; "
               "1.40x = MFMA count)\n", d / w, d / pp, d / l);
    }
    return 0;
}
