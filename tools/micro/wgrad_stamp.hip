// Diagnostic: the brick weight-gradient kernel (tdx_conv3_wgrad_mfma.hip built with W3_STAMPS), standalone: s_memtime
// stamps of 8 brick iterations of every wave -> where an iteration's cycles go.  The kernel is the product kernel (merge
// included), plus stamps.  Build: hipcc -O3 --offload-arch=gfx950 wgrad_stamp.hip -o wgrad_stamp
// Run: ./wgrad_stamp [Cin] [Cout] [X Y Z]
#define W3_STAMPS 1
#include "../../generative-turbulence_amd/csrc/tdx_conv3_wgrad_mfma.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
int conv3_wgrad_small_launch(const void*, int, const void*, int, const void*, float*, float*, int, int, int, int, int, hipStream_t, float*, int, int*) { return TDX_ESHAPE; }

__global__ void fill_rand(unsigned* p, size_t n, unsigned seed, unsigned expo) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned a = (h & 0x83ff) + expo, b = ((h >> 16) & 0x83ff) + expo;
        p[i] = (a & 0xffff) | (b << 16);
    }
}

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 64, Cout = argc > 2 ? atoi(argv[2]) : 64;
    const int X = argc > 5 ? atoi(argv[3]) : 192, Y = argc > 5 ? atoi(argv[4]) : 64, Z = argc > 5 ? atoi(argv[5]) : 48;
    const int B = 6;
    const size_t nx = (size_t)B * X * Y * Z * Cin, ny = (size_t)B * X * Y * Z * Cout;
    bf16 *x, *dy; float *dw, *db;
    (void)hipMalloc(&x, nx * 2); (void)hipMalloc(&dy, ny * 2); (void)hipMalloc(&dw, (size_t)27 * Cin * Cout * 4 + 4096); (void)hipMalloc(&db, 4096);
    fill_rand<<<2048, 256>>>((unsigned*)x, nx / 2, 1, 0x3c00);
    fill_rand<<<2048, 256>>>((unsigned*)dy, ny / 2, 2, 0x3c00);
    const size_t nrec = (size_t)1024 * 4 * (W3_NSTAMP + 1);
    unsigned long long* buf;
    (void)hipMalloc(&buf, nrec * 8);
    (void)hipMemset(buf, 0, nrec * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(w3_stamps_dev), &buf, sizeof(buf));
    auto go = [&]() { return conv3_wgrad_mfma_launch(x, Cin, nullptr, 0, dy, dw, db, B, X, Y, Z, Cout, 0, nullptr, 0, nullptr); };
    int rc = go();
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    for (int it = 0; it < 20; ++it) go();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) go();
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("# wgrad brick kernel Cin %d Cout %d grid %dx%dx%d B %d: %.3f ms/launch, %.0f TFLOP/s (with stamps)\n", Cin, Cout, X, Y, Z, B, ms,
           54.0 * Cin * Cout * B * X * Y * Z / ms / 1e9);
    std::vector<unsigned long long> h(nrec);
    (void)hipMemcpy(h.data(), buf, nrec * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"barrier 1 (+ wait for the staged loads)", "LDS stores", "barrier 2", "next brick's loads issued", "MFMA phase", "to next iteration"};
    double sum[6] = {0, 0, 0, 0, 0, 0};
    size_t n = 0;
    for (size_t wv = 0; wv < (size_t)1024 * 4; ++wv) {
        const unsigned long long* rec = &h[wv * (W3_NSTAMP + 1)];
        if (rec[0] < 48) continue;
        for (int it = 0; it < 7; ++it) {
            const unsigned long long* q = rec + 1 + 6 * it;
            for (int k = 0; k < 5; ++k) sum[k] += (double)(q[k + 1] - q[k]);
            sum[5] += (double)(q[6] - q[5]);
            ++n;
        }
    }
    if (!n) { printf("no stamps\n"); return 1; }
    double tot = 0;
    for (int k = 0; k < 6; ++k) tot += sum[k] / n;
    for (int k = 0; k < 6; ++k) printf("#   %-44s %7.0f cycles (%4.1f %%)\n", names[k], sum[k] / n, 100 * sum[k] / n / tot);
    printf("#   iteration %.0f cycles; MFMA issue time at 32 cycles each: %d\n", tot, (Cout % 64 == 0 ? 224 : 112) * 32);
    return 0;
}
