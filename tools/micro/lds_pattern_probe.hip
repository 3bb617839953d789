// Which 16-B slot patterns does a ds_read_b128 serve without bank conflicts on gfx950?  Four waves per SIMD issue 4096 reads each with a
// given lane -> entry mapping (entry = 16-B slot index; constant offsets added per read, as the conv kernels' tap offsets are);
// time per read against the linear pattern (lane i -> slot i) tells the serialisation factor.
// Patterns: the ring conv kernel's fragment reads -- 8-deep bricks (z stride 12), 4-deep bricks (z stride 6, round 6), and candidates.
// Build: hipcc -O3 --offload-arch=gfx950 lds_pattern_probe.hip -o lds_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <algorithm>

__global__ void __launch_bounds__(1024) probe(const int* entry, unsigned* out, unsigned long long* cycles, int iters) {
    __shared__ uint4 sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) sm[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int e = entry[lane];
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint4 v = sm[(e + 37 * k + it + 64 * (threadIdx.x >> 6)) & 8191];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) out[0] = acc.y ^ acc.z ^ acc.w;
}

static double run(const char* name, const int* h_entry) {
    int* d_entry; unsigned* out; unsigned long long* cyc;
    (void)hipMalloc(&d_entry, 64 * 4); (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8 * 256);
    (void)hipMemcpy(d_entry, h_entry, 64 * 4, hipMemcpyHostToDevice);
    const int iters = 256;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 0, 0, d_entry, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double per = s / 256 / (iters * 16.0);
    printf("%-64s %7.2f memtime ticks per ds_read_b128 (16 waves per CU: throughput-bound)\n", name, per);
    (void)hipFree(d_entry); (void)hipFree(out); (void)hipFree(cyc);
    return per;
}

int main() {
    int e[64];
    auto both = [&](auto f) { for (int l = 0; l < 64; ++l) { const int r = l & 31, hh = l >> 5; e[l] = f(r) + hh * 1; } };
    for (int l = 0; l < 64; ++l) e[l] = l;
    const double lin = run("linear: lane i -> slot i", e);
    for (int l = 0; l < 64; ++l) e[l] = 16 * l;
    run("worst: every lane in the same 16-B bank group (stride 16)", e);
    both([](int r) { return 12 * (r & 3) + (r >> 2); });
    run("ring, 8-deep bricks: 12 (r & 3) + (r >> 2), halves one tap apart", e);
    both([](int r) { return 6 * ((r >> 1) & 7) + (r & 1) + 2 * (r >> 4); });
    run("ring, 4-deep bricks (round 6): 6 ((r>>1)&7) + (r&1) + 2 (r>>4)", e);
    both([](int r) { return 6 * (r & 7) + (r >> 3); });
    run("candidate: 6 (r & 7) + (r >> 3)", e);
    both([](int r) { return 12 * (r & 7) + (r >> 3); });
    run("candidate: z stride 12, 12 (r & 7) + (r >> 3)", e);
    both([](int r) { return 6 * (r >> 2) + (r & 3); });
    run("candidate: 6 (r >> 2) + (r & 3)  (y = r >> 2, z = r & 3)", e);
    both([](int r) { return 8 * (r >> 2) + (r & 3); });
    run("candidate: z stride 8, 8 (r >> 2) + (r & 3)", e);
    both([](int r) { return 8 * (r & 7) + (r >> 3); });
    run("candidate: z stride 8, 8 (r & 7) + (r >> 3)", e);
    // halves further apart (a tap pair whose second tap sits a y row below: + z stride)
    for (int l = 0; l < 64; ++l) { const int r = l & 31, hh = l >> 5; e[l] = 6 * ((r >> 1) & 7) + (r & 1) + 2 * (r >> 4) + hh * 6; }
    run("4-deep bricks, halves one y row apart (+6)", e);
    for (int l = 0; l < 64; ++l) { const int r = l & 31, hh = l >> 5; e[l] = 12 * (r & 3) + (r >> 2) + hh * 12; }
    run("8-deep bricks, halves one y row apart (+12)", e);
    // the small-grid conv kernel (tdx_conv3_small_kernel.h): lane r = row r of a densely packed M tile, image with a one-voxel rim
    for (int l = 0; l < 64; ++l) { const int r = l & 31; e[l] = (r / 6) * 8 + r % 6 + (l >> 5) * 4096; }  // halves = planes IMG_HALF apart
    run("small-grid kernel, 24 x 8 x 6 level (Iz = 8): (r / 6) 8 + r % 6", e);
    for (int l = 0; l < 64; ++l) { const int r = l & 31; e[l] = ((r / 12) * 6 + (r / 3) % 4) * 5 + r % 3 + (l >> 5) * 4096; }
    run("small-grid kernel, 12 x 4 x 3 level (Iy = 6, Iz = 5)", e);
    for (int l = 0; l < 64; ++l) { const int r = l & 31; e[l] = ((r / 80) * 10 + (r / 8) % 10) * 10 + r % 8 + (l >> 5) * 4096; }
    run("small-grid kernel, data gradient on the padded 26 x 10 x 8 grid (Iz = 10)", e);
    printf("(linear = %.2f)\n", lin);
    // search: 32 lanes <-> (y in 0..7, z in 0..3) by every permutation of the five lane-index bits, image z stride 6 .. 11;
    // entry = stride_y y + z with stride_y = the z stride (rows of the halo'd 4-deep brick are contiguous in z)
    int perm[5] = {0, 1, 2, 3, 4};
    double best[12];
    int best_perm[12][5];
    for (int sz = 6; sz < 12; ++sz) best[sz] = 1e30;
    do {
        for (int sz = 6; sz < 12; ++sz) {
            for (int l = 0; l < 64; ++l) {
                const int r = l & 31;
                int v = 0;  // v bit i = r bit perm[i]; y = v & 7, z = v >> 3
                for (int i = 0; i < 5; ++i) v |= ((r >> perm[i]) & 1) << i;
                e[l] = sz * (v & 7) + (v >> 3) + (l >> 5);
            }
            int* d_entry; unsigned* out; unsigned long long* cyc;
            (void)hipMalloc(&d_entry, 256); (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8 * 256);
            (void)hipMemcpy(d_entry, e, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(64), dim3(1024), 0, 0, d_entry, out, cyc, 64);
            unsigned long long h[64];
            (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < 64; ++i) sum += (double)h[i];
            const double per = sum / 64 / (64 * 16.0);
            if (per < best[sz]) { best[sz] = per; memcpy(best_perm[sz], perm, sizeof(perm)); }
            (void)hipFree(d_entry); (void)hipFree(out); (void)hipFree(cyc);
        }
    } while (std::next_permutation(perm, perm + 5));
    // the small-grid kernel's rows (lane r = voxel r of a (y, z)-ordered run, Z voxels per z row) for image z strides Z + 2 .. 16
    // and y strides Iy * sz (+ 0 .. 3 entries of padding per x plane): what would a padded image buy?
    for (int Z : {3, 5, 6, 8}) {
        const int Y = Z == 3 ? 4 : (Z == 5 ? 6 : (Z == 6 ? 8 : 10));
        printf("rows of %d voxels (Y = %d):", Z, Y);
        for (int sz = Z + 2; sz <= 16; ++sz) {
            double worst = 0;
            for (int start = 0; start < Z * Y; start += 5) {  // M tiles start anywhere in the run
                for (int l = 0; l < 64; ++l) {
                    const int v = start + (l & 31), lx = v / (Z * Y), ly = (v / Z) % Y, lz = v % Z;
                    e[l] = (lx * (Y + 2) + ly) * sz + lz + (l >> 5) * 4096;
                }
                int* d_entry; unsigned* out; unsigned long long* cyc;
                (void)hipMalloc(&d_entry, 256); (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8 * 256);
                (void)hipMemcpy(d_entry, e, 256, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(probe, dim3(64), dim3(1024), 0, 0, d_entry, out, cyc, 64);
                unsigned long long h[64];
                (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                double sum = 0;
                for (int i = 0; i < 64; ++i) sum += (double)h[i];
                worst = std::max(worst, sum / 64 / (64 * 16.0));
                (void)hipFree(d_entry); (void)hipFree(out); (void)hipFree(cyc);
            }
            printf(" sz %d: %.0f", sz, worst);
        }
        printf("\n");
    }
    for (int sz = 6; sz < 12; ++sz)
        printf("z stride %2d: best %.2f ticks with v bits <- lane bits {%d %d %d %d %d} (y = v & 7, z = v >> 3)\n", sz, best[sz], best_perm[sz][0],
               best_perm[sz][1], best_perm[sz][2], best_perm[sz][3], best_perm[sz][4]);
    return 0;
}
