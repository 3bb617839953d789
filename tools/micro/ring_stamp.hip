// Diagnostic: the ring conv kernel (tdx_conv3_ring.hip) built with RG_STAMPS, standalone (no torch): per-unit
// s_memtime stamps of two bricks of every wave -> where a unit's cycles go (wait for copies / barrier / MFMA + DMA
// issue) and what the epilogue costs.  Build: hipcc -O3 --offload-arch=gfx950 ring_stamp.hip -o ring_stamp
// Run: ./ring_stamp [Cin] [Cout] [X Y Z] [zero_pad]
#define RG_STAMPS 1
#define RG_STAMP_B0 3
#include "../../generative-turbulence_amd/csrc/tdx_conv3_ring.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

static void* g_scratch = nullptr;
void* tdx_scratch_ptr() { return g_scratch; }
size_t tdx_scratch_bytes() { return 1 << 20; }
bool conv3_mfma_supported(int C1, int C2, int Cout) { return C1 > 0 && (C1 % 16) == 0 && (C2 % 16) == 0 && (Cout % 32) == 0; }

__global__ void fill_rand(unsigned* p, size_t n, unsigned seed, unsigned expo) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // two bf16 with random sign, mantissa and 3 exponent bits (|v| in [2^-7 .. 2) * scale): real-data-like switching
        const unsigned a = (h & 0x83ff) + expo, b = ((h >> 16) & 0x83ff) + expo;
        p[i] = (a & 0xffff) | (b << 16);
    }
}

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 64, Cout = argc > 2 ? atoi(argv[2]) : 64;
    const int X = argc > 5 ? atoi(argv[3]) : 192, Y = argc > 5 ? atoi(argv[4]) : 64, Z = argc > 5 ? atoi(argv[5]) : 48;
    const bool zp = argc > 6 && atoi(argv[6]);
    const int B = 6;
    const size_t nx = (size_t)B * X * Y * Z * Cin, nw = (size_t)27 * Cin * Cout, ny = (size_t)B * X * Y * Z * Cout;
    bf16 *x, *w, *y;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2);
    hipMalloc(&g_scratch, 1 << 20); hipMemset(g_scratch, 0, 1 << 20);
    fill_rand<<<2048, 256>>>((unsigned*)x, nx / 2, 1, 0x3c00);
    fill_rand<<<256, 256>>>((unsigned*)w, nw / 2, 2, 0x3800);
    const size_t nrec = (size_t)256 * RG_WAVES * (RG_NSTAMP + 1);
    hipMalloc(&rg_stamp_buffer, nrec * 8);
    hipMemset(rg_stamp_buffer, 0, nrec * 8);
    double* gn; hipMalloc(&gn, (size_t)32 * B * Cout * 2 * 8); hipMemset(gn, 0, (size_t)32 * B * Cout * 2 * 8);
    auto go = [&]() {
        return zp ? conv3_ring_launch(x, Cin, nullptr, 0, w, nullptr, nullptr, B, X, Y, Z, Cout, true, 0, nullptr, y, Cout, nullptr, nullptr, nullptr)
                  : conv3_ring_launch(x, Cin, nullptr, 0, w, nullptr, y, B, X, Y, Z, Cout, false, 0, gn);
    };
    int rc = go();
    if (rc != 0) { printf("launch failed: %d\n", rc); return 1; }
    for (int it = 0; it < 40; ++it) go();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) go();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double fl = 54.0 * Cin * Cout * B * X * Y * Z;
    printf("# ring kernel Cin %d Cout %d grid %dx%dx%d B %d zero_pad %d: %.3f ms/launch, %.0f TFLOP/s (with stamps)\n", Cin, Cout, X, Y, Z, B,
           (int)zp, ms, fl / ms / 1e9);
    std::vector<unsigned long long> h(nrec);
    hipMemcpy(h.data(), rg_stamp_buffer, nrec * 8, hipMemcpyDeviceToHost);
    // per brick: nun units x 3 stamps (top, own copies landed, barrier passed), then "last MFMA issued", "stores issued"
    const int nun = Cin / 8, per_brick = 3 * nun + 2;
    const int nbk = 2 * per_brick <= RG_NSTAMP ? 2 : 1;
    double wait = 0, bar = 0, body = 0, epi = 0, gap = 0, brick = 0;
    std::vector<long long> bodies, waits, bars;
    size_t n = 0, nu = 0;
    for (size_t wv = 0; wv < (size_t)256 * RG_WAVES; ++wv) {
        const unsigned long long* rec = &h[wv * (RG_NSTAMP + 1)];
        if ((int)rec[0] < nbk * per_brick) continue;
        const unsigned long long* s = rec + 1;
        for (int bk = 0; bk < nbk; ++bk) {
            const unsigned long long* q = s + bk * per_brick;
            for (int u = 0; u < nun; ++u) {
                const unsigned long long top = q[3 * u], landed = q[3 * u + 1], passed = q[3 * u + 2];
                const unsigned long long next = u + 1 < nun ? q[3 * u + 3] : q[3 * nun];
                wait += (double)(landed - top); bar += (double)(passed - landed); body += (double)(next - passed);
                waits.push_back((long long)(landed - top)); bars.push_back((long long)(passed - landed)); bodies.push_back((long long)(next - passed));
                ++nu;
            }
            epi += (double)(q[3 * nun + 1] - q[3 * nun]);
            if (bk == 0 && nbk == 2) { gap += (double)(q[per_brick] - q[3 * nun + 1]); brick += (double)(q[per_brick] - q[0]); }
        }
        ++n;
    }
    if (!n) { printf("no stamps\n"); return 1; }
    auto pct = [](std::vector<long long>& v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("# %zu waves; per unit = 8 channels x 28 taps (shader cycles; MFMA time of a unit = %d per wave, x2 for the two waves of a SIMD):\n", n, 14 * 4 * 32);
    printf("#   wait for own copies   mean %.0f  p50 %lld p90 %lld p99 %lld\n", wait / nu, pct(waits, .5), pct(waits, .9), pct(waits, .99));
    printf("#   barrier               mean %.0f  p50 %lld p90 %lld p99 %lld\n", bar / nu, pct(bars, .5), pct(bars, .9), pct(bars, .99));
    printf("#   body (MFMA + DMA)     mean %.0f  p50 %lld p90 %lld p99 %lld\n", body / nu, pct(bodies, .5), pct(bodies, .9), pct(bodies, .99));
    printf("# per brick: epilogue (tiles, stores, stats) %.0f, to next unit top %.0f; brick period %.0f cycles\n", epi / (nbk * n), gap / n, brick / n);
    // one wave's raw timeline
    const unsigned long long* rec = &h[0];
    printf("wave 0 of workgroup 0:");
    for (int i = 0; i < per_brick + 3; ++i) printf(" %llu", rec[1 + i] - rec[1]);
    printf("\n");
    return 0;
}
