// Per-kernel boundary cost when the GPU is never waiting for the host: 300 dependent kernels of ~20 us each (256 workgroups
// spinning on the wall clock), enqueued eagerly in one stream (the host is far ahead after the first few) vs replayed as one
// captured hipGraph.  (total - 300 x spin) / 300 = what one kernel boundary costs on the device in each mode -- the question
// behind "the captured B = 6 training step (~700 nodes) is 0.5 ms slower than the eager one".
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) spin(long long cycles, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < cycles) ++v;
    if (v == 0xffffffffu) sink[0] = v;
}

int main() {
    unsigned* sink; (void)hipMalloc(&sink, 4);
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 300;
    for (int us : {5, 20, 60}) {
        const long long cyc = (long long)rate * us / 1000;
        for (int W : {64, 256, 1024}) {
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(W), dim3(256), 0, st, cyc, sink);
            (void)hipStreamEndCapture(st, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            float eb = 1e9f, gb = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, st);
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(W), dim3(256), 0, st, cyc, sink);
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < eb) eb = ms;
                (void)hipEventRecord(e0, st);
                (void)hipGraphLaunch(ge, st);
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1); if (ms < gb) gb = ms;
            }
            printf("%2d-us kernels of %4d workgroups: eager %.2f us per kernel (boundary %.2f), graph %.2f us per node (boundary %.2f)\n", us, W,
                   1e3f * eb / N, 1e3f * eb / N - us, 1e3f * gb / N, 1e3f * gb / N - us);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
