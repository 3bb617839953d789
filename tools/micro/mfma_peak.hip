// Sustained dense bf16 MFMA rate of this chip with no memory traffic at all: the practical ceiling
// for any MFMA kernel (power / clock management included).  Build: hipcc -O3 --offload-arch=gfx950
// mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
    if (s == 12345.678f) out[0] = s;
}

// short workgroups, as the conv kernel launches them: `iters` x 4 MFMAs per wave, optional LDS
// allocation (dynamic), optional prologue of NLOAD 16-B global loads per thread staged to LDS
template <int NLOAD>
__global__ void __launch_bounds__(256, 2) kshort(float* out, int iters, const uint4* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    if (NLOAD > 0) {
        uint4 r[NLOAD > 0 ? NLOAD : 1];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) r[i] = src[(size_t)(blockIdx.x % 4096) * 256 * NLOAD + i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) *reinterpret_cast<uint4*>(smem + (i * 256 + threadIdx.x) * 16) = r[i];
        __syncthreads();
        a = *reinterpret_cast<bf16x8*>(smem + ((threadIdx.x * 7) % (256 * NLOAD)) * 16);
    }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 12345.678f) out[0] = s;
}
// operands with random bit patterns (switching activity of real data), NSET register sets rotating
template <int NSET>
__global__ void __launch_bounds__(256, 2) krand(float* out, int iters, const uint4* src) {
    bf16x8 a[NSET], b[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        uint4 ua = src[(s * 2) * 256 + threadIdx.x], ub = src[(s * 2 + 1) * 256 + threadIdx.x];
        a[s] = *reinterpret_cast<bf16x8*>(&ua); b[s] = *reinterpret_cast<bf16x8*>(&ub);
    }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[(s + (n >> 1)) % NSET], acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    if (s == 12345.678f) out[0] = s;
}
template <int NSET>
static void run_rand(const char* name, int blocks, int iters, int reps, bool random) {
    float* out; hipMalloc(&out, 4);
    const size_t n = 16 * 256 * 16;
    unsigned short* h = (unsigned short*)malloc(n);
    for (size_t i = 0; i < n / 2; ++i) {
        // bf16 normal-ish values in (-2, 2): random sign, exponent 120..127, random mantissa
        h[i] = random ? (unsigned short)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127)) : (unsigned short)0x3f80;
    }
    uint4* src; hipMalloc(&src, n); hipMemcpy(src, h, n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(krand<NSET>, dim3(blocks), dim3(256), 0, 0, out, iters, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(krand<NSET>, dim3(blocks), dim3(256), 0, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 4.0 * NSET * 32 * 32 * 16 * 2 * iters * 4.0 * blocks * reps;
    printf("%-60s blocks %5d iters %4d: %8.3f ms/launch, %7.1f TFLOP/s\n", name, blocks, iters, ms / reps, flops / ms / 1e9);
    hipFree(out); hipFree(src); free(h);
}

template <int NLOAD>
static void run_short(const char* name, int blocks, int iters, size_t lds, int reps) {
    float* out; hipMalloc(&out, 4);
    uint4* src; hipMalloc(&src, (size_t)4096 * 256 * 20 * 16); hipMemset(src, 0, (size_t)4096 * 256 * 20 * 16);
    auto kern = kshort<NLOAD>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 4.0 * 32 * 32 * 16 * 2 * iters * 4.0 * blocks * reps;
    printf("%-60s blocks %5d iters %4d lds %6zu: %8.3f ms/launch, %7.1f TFLOP/s\n", name, blocks, iters, lds, ms / reps, flops / ms / 1e9);
    hipFree(out); hipFree(src);
}

template <typename K>
static void run(const char* name, K kern, int blocks, int iters, double flop_per_iter_per_wave, int reps) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = flop_per_iter_per_wave * iters * 4.0 * blocks * reps;
    printf("%-44s blocks %5d iters %6d reps %3d: %8.3f ms total, %7.1f TFLOP/s\n", name, blocks, iters, reps, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const double f32 = 32.0 * 32 * 16 * 2, f16 = 16.0 * 16 * 32 * 2;
    // short bursts (a conv-launch-like 0.3 ms) and sustained (tens of ms)
    for (int reps : {1, 100}) {
        run("32x32x16 4 acc, 1 wave/SIMD (256 WG)", k32<4>, 256, 4000, 4 * f32, reps);
        run("32x32x16 4 acc, 2 waves/SIMD (512 WG)", k32<4>, 512, 2000, 4 * f32, reps);
        run("32x32x16 8 acc, 2 waves/SIMD", k32<8>, 512, 1000, 8 * f32, reps);
        run("32x32x16 2 acc, 2 waves/SIMD", k32<2>, 512, 4000, 2 * f32, reps);
        run("32x32x16 1 acc (dependent), 2 waves/SIMD", k32<1>, 512, 8000, 1 * f32, reps);
        run("32x32x16 1 acc (dependent), 1 wave/SIMD", k32<1>, 256, 16000, 1 * f32, reps);
        run("32x32x16 2 acc, 1 wave/SIMD", k32<2>, 256, 8000, 2 * f32, reps);
        run("16x16x32 4 acc, 2 waves/SIMD", k16<4>, 512, 4000, 4 * f16, reps);
        run("16x16x32 8 acc, 2 waves/SIMD", k16<8>, 512, 2000, 8 * f16, reps);
        run("32x32x16 4 acc, 4 waves/SIMD (1024 WG)", k32<4>, 1024, 1000, 4 * f32, reps);
    }
    run_short<0>("short WG, no LDS", 13824, 108, 0, 20);
    run_short<0>("short WG, 78 KB LDS (2 WG/CU)", 13824, 108, 78 * 1024, 20);
    run_short<0>("short WG, 40 KB LDS (4 WG/CU by LDS, 2 by launch bounds)", 13824, 108, 40 * 1024, 20);
    run_short<20>("short WG, 78 KB LDS + 20 global loads/thread prologue", 13824, 108, 78 * 1024, 20);
    run_short<6>("short WG, 78 KB LDS + 6 global loads/thread prologue", 13824, 108, 78 * 1024, 20);
    run_short<0>("4x longer WG, 78 KB LDS", 13824 / 4, 432, 78 * 1024, 20);
    run_short<20>("4x longer WG, 78 KB LDS + 20 loads", 13824 / 4, 432, 78 * 1024, 20);
    run_short<0>("persistent-size WG (512), 78 KB LDS", 512, 2916, 78 * 1024, 20);
    run_rand<1>("all-ones operands, 1 set", 512, 2000, 100, false);
    run_rand<1>("random operands, 1 register set", 512, 2000, 100, true);
    run_rand<4>("random operands, 4 register sets rotating", 512, 500, 100, true);
    run_rand<4>("random operands, 4 sets, 1 wave/SIMD", 256, 1000, 100, true);
    run_rand<4>("random operands, 4 sets, short burst", 512, 500, 1, true);
    return 0;
}
