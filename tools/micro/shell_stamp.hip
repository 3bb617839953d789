// Diagnostic: the halo-shell kernel of the data gradient (tdx_conv3_shell.hip built with SH_STAMPS), standalone: s_memtime
// stamps of every wave's phases -> where a workgroup's life goes, and how the workgroups' lives tile the launch.
// Build: hipcc -O3 --offload-arch=gfx950 shell_stamp.hip -o shell_stamp      Run: ./shell_stamp [K] [N] [X Y Z]
#define SH_STAMPS 1
#include "../../generative-turbulence_amd/csrc/tdx_conv3_shell.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

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
    const int K = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 64;
    const int X = argc > 5 ? atoi(argv[3]) : 192, Y = argc > 5 ? atoi(argv[4]) : 64, Z = argc > 5 ? atoi(argv[5]) : 48;
    const int B = 6;
    const size_t ndy = (size_t)B * X * Y * Z * K, ndx = (size_t)B * X * Y * Z * N;
    bf16 *dy, *dx, *wb;
    (void)hipMalloc(&dy, ndy * 2); (void)hipMalloc(&dx, ndx * 2); (void)hipMalloc(&wb, (size_t)27 * K * N * 2);
    fill_rand<<<2048, 256>>>((unsigned*)dy, ndy / 2, 1, 0x3c00);
    fill_rand<<<64, 256>>>((unsigned*)wb, (size_t)27 * K * N / 2, 2, 0x3800);
    (void)hipMemset(dx, 0, ndx * 2);
    const size_t maxwg = 65536, nrec = maxwg * 4 * SH_NSTAMP;
    unsigned long long* buf;
    (void)hipMalloc(&buf, nrec * 8);
    (void)hipMemset(buf, 0, nrec * 8);
    unsigned long long* null = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sh_stamps_dev), &null, sizeof(null));
    auto go = [&]() { return conv3_shell_launch(dy, wb, dx, N, nullptr, B, X, Y, Z, K, N, SH_BF16, nullptr); };
    int rc = go();
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    for (int it = 0; it < 20; ++it) go();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) go();
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double pos = 2.0 * ((X + 2.0) * (Y + 2) + (X + 2.0) * Z + (double)Y * Z) * B;
    printf("# shell kernel K %d -> N %d grid %dx%dx%d B %d: %.1f us/launch (stamps off), %.0f TFLOP/s over %.0f shell positions\n", K, N, X, Y,
           Z, B, ms * 1e3, 18.0 * K * N * pos / ms / 1e9, pos);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sh_stamps_dev), &buf, sizeof(buf));
    go(); (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(nrec);
    (void)hipMemcpy(h.data(), buf, nrec * 8, hipMemcpyDeviceToHost);
    const int kc = 32;  // S = 2 slices of 16 channels per chunk where K divides
    const int nchunks = K % kc == 0 ? K / kc : K / 16;
    // stamp order: 0 start, 1 plans, then per chunk c < 2: slice in LDS, MFMAs issued; then K loop done, tile in LDS, fold done
    const int per = 2 + 2 * std::min(nchunks, 2) + 3;
    std::vector<double> sum(per, 0.0);
    unsigned long long t0 = ~0ull, t1 = 0;
    size_t n = 0;
    std::vector<double> life;
    for (size_t wv = 0; wv < maxwg * 4; ++wv) {
        const unsigned long long* q = &h[wv * SH_NSTAMP];
        if (q[0] == 0) continue;
        for (int k = 0; k + 1 < per; ++k) sum[k] += (double)(q[k + 1] - q[k]);
        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[per - 1]);
        life.push_back((double)(q[per - 1] - q[0]));
        ++n;
    }
    if (!n) { printf("no stamps\n"); return 1; }
    std::vector<const char*> names = {"block -> patch, staging plans"};
    names.push_back("first slice: global loads -> LDS (2 barriers)");
    names.push_back("MFMAs of chunk 0 (+ loads of chunk 1 issued)");
    if (nchunks >= 2) { names.push_back("chunk 1: wait + LDS stores (2 barriers)"); names.push_back("MFMAs of chunk 1"); }
    names.push_back(nchunks > 2 ? "chunks 2 .. end" : "K loop exit");
    names.push_back("accumulators -> LDS tile (2 barriers)");
    names.push_back("fold: read-add-write / atomics on dx");
    double tot = 0;
    for (int k = 0; k + 1 < per; ++k) tot += sum[k] / n;
    for (int k = 0; k + 1 < per; ++k) printf("#   %-50s %7.0f cycles (%4.1f %%)\n", names[k], sum[k] / n, 100 * sum[k] / n / tot);
    std::sort(life.begin(), life.end());
    printf("#   wave life %.0f cycles (median %.0f, max %.0f) (s_memtime = shader clock); launch span %.0f cycles = %.2f x the kernel time at 2 GHz; %zu waves = %zu workgroups\n",
           tot, life[life.size() / 2], life.back(), (double)(t1 - t0), (double)(t1 - t0) / 2000.0 / (ms * 1e3), n, n / 4);
    return 0;
}
