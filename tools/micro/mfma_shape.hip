// Bare bf16 MFMA loops on RANDOM operands re-read from LDS: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16, same
// FLOPs per wave and the same LDS bytes per FLOP (a 64 x 64 tile per wave and K = 32 per step: 4 + 4 fragment reads).
// Question: in a power-limited loop, does the 16x16x32 shape hold a higher clock (MI355X_MICROARCH.md, DVFS item 7)?
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool LDS_READS>
__global__ void __launch_bounds__(512, 2) k32(float* out, const uint4* src, int iters) {
    __shared__ uint4 sm[4096];  // 64 KiB of random operand data
    for (int i = threadIdx.x; i < 4096; i += 512) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2][2];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) b[i] = 0.f;
    bf16x8 A[2], B[2];
    int off = (wave * 64 + lane) & 4095;
    for (int j = 0; j < 2; ++j) { A[j] = __builtin_bit_cast(bf16x8, sm[(off + 64 * j) & 4095]); B[j] = __builtin_bit_cast(bf16x8, sm[(off + 512 + 64 * j) & 4095]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two K = 16 steps = K 32
            if (LDS_READS) {
                off = (off + 257) & 4095;
                for (int j = 0; j < 2; ++j) { A[j] = __builtin_bit_cast(bf16x8, sm[(off + 64 * j) & 4095]); B[j] = __builtin_bit_cast(bf16x8, sm[(off + 2048 + 64 * j) & 4095]); }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[n], acc[m][n], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}

template <bool LDS_READS>
__global__ void __launch_bounds__(512, 2) k16(float* out, const uint4* src, int iters) {
    __shared__ uint4 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4][4];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 4; ++i) b[i] = 0.f;
    bf16x8 A[4], B[4];
    int off = (wave * 64 + lane) & 4095;
    for (int j = 0; j < 4; ++j) { A[j] = __builtin_bit_cast(bf16x8, sm[(off + 64 * j) & 4095]); B[j] = __builtin_bit_cast(bf16x8, sm[(off + 512 + 64 * j) & 4095]); }
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {  // one K = 32 step: 4 + 4 fragments
            off = (off + 257) & 4095;
            for (int j = 0; j < 4; ++j) { A[j] = __builtin_bit_cast(bf16x8, sm[(off + 64 * j) & 4095]); B[j] = __builtin_bit_cast(bf16x8, sm[(off + 2048 + 64 * j) & 4095]); }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m], B[n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 4; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}

// 32x32x16 with an MT x NT wave tile (MT x 32 voxels by NT x 32 channels), WAVES waves per workgroup, one workgroup per CU:
// LDS bytes per MFMA = (MT + NT) / (MT NT) KiB -- 1 for 2 x 2, 0.75 for 4 x 2, 0.5 for 4 x 4
template <int MT, int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) ktile(float* out, const uint4* src, int iters) {
    __shared__ uint4 sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) sm[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[MT][NT];
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) b[i] = 0.f;
    bf16x8 A[MT], B[NT];
    int off = (wave * 64 + lane) & 4095;
    for (int it = 0; it < iters; ++it) {
        off = (off + 257) & 4095;
#pragma unroll
        for (int j = 0; j < MT; ++j) A[j] = __builtin_bit_cast(bf16x8, sm[(off + 64 * j) & 4095]);
#pragma unroll
        for (int j = 0; j < NT; ++j) B[j] = __builtin_bit_cast(bf16x8, sm[(off + 2048 + 64 * j) & 4095]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[m], B[n], acc[m][n], 0, 0, 0);
    }
    float s = 0.f;
    for (auto& a : acc) for (auto& b : a) for (int i = 0; i < 16; ++i) s += b[i];
    if (s == 12345.678f) out[0] = s;
}
template <int MT, int NT, int WAVES>
static void run_tile(const char* name, const uint4* src, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = ktile<MT, NT, WAVES>;
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), 0, 0, out, src, iters);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), 0, 0, out, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * (32.0 * MT) * (32.0 * NT) * 16 * iters * WAVES * 256.0 * reps;
    printf("%-44s %8.3f ms/launch  %7.1f TFLOP/s\n", name, ms / reps, flops / ms / 1e9);
    (void)hipFree(out);
}

template <typename K>
static void run(const char* name, K kern, const uint4* src, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 30; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, src, iters);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, src, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 64 * 64 * 32 * iters * 8.0 * 256 * reps;  // 64 x 64 x 32 per wave and iteration
    printf("%-44s %8.3f ms/launch  %7.1f TFLOP/s\n", name, ms / reps, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    const size_t n = 4096 * 16;
    unsigned short* h = (unsigned short*)malloc(n);
    srand(1);
    for (size_t i = 0; i < n / 2; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127));
    uint4* src; (void)hipMalloc(&src, n); (void)hipMemcpy(src, h, n, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run("32x32x16, operands in registers", k32<false>, src, 4000);
        run("16x16x32, operands in registers", k16<false>, src, 4000);
        run("32x32x16, operands re-read from LDS", k32<true>, src, 4000);
        run("16x16x32, operands re-read from LDS", k16<true>, src, 4000);
        run_tile<2, 2, 8>("32x32x16 LDS-fed, 64x64 tile, 8 waves", src, 8000);
        run_tile<2, 2, 4>("32x32x16 LDS-fed, 64x64 tile, 4 waves", src, 16000);
        run_tile<4, 2, 4>("32x32x16 LDS-fed, 128x64 tile, 4 waves", src, 8000);
        run_tile<4, 2, 8>("32x32x16 LDS-fed, 128x64 tile, 8 waves", src, 4000);
        run_tile<4, 4, 4>("32x32x16 LDS-fed, 128x128 tile, 4 waves", src, 4000);
    }
    return 0;
}
