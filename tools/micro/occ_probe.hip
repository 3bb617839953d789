// How many workgroups of 256 threads with L bytes of dynamic LDS are co-resident per CU?
// 512 (= 2 per CU) / 768 workgroups that each spin ~200 us: elapsed = 200 us x ceil(WGs per CU / resident).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 2) spin(long long cycles, int* out) {
    extern __shared__ unsigned char smem[];
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(10);
    if (cycles < 0) out[0] = smem[threadIdx.x];
}
int main() {
    int* out; (void)hipMalloc(&out, 4);
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    for (size_t lds : {0ul, 40000ul, 65536ul, 78592ul, 81920ul, 100000ul}) {
        (void)hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int blocks : {256, 512, 768}) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), lds, 0, (long long)rate / 5, out);  // 0.2 ms
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), lds, 0, (long long)rate / 5, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("lds %6zu B, %3d workgroups: %.3f ms (%s)\n", lds, blocks, ms, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
