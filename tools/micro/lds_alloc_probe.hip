// Which bits of HW_REG_LDS_ALLOC tell two co-resident workgroups apart?  (gfx950; 2 workgroups of 80 KB LDS per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    unsigned* d; hipMalloc(&d, 1024 * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80640);
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), 80640, 0, d);
    unsigned h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 512; i += 37) printf("wg %3d lds_alloc %08x hw_id %08x\n", i, h[2 * i], h[2 * i + 1]);
    int n0 = 0; for (int i = 0; i < 512; ++i) n0 += (h[2 * i] & 0xfff) == 0;
    printf("workgroups with base field (low 12 bits) == 0: %d of 512\n", n0);
    return 0;
}
