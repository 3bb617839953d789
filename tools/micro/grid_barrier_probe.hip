// What does a grid-wide barrier cost inside one persistent launch, against what a kernel boundary costs inside a replayed
// hipGraph?  (B = 1 sampling runs ~50 launches of 4-40 us at 24x8x6 and below: a cooperative kernel for the deep levels
// pays one barrier where the graph pays one node boundary.)
//   barrier: W workgroups x 256 threads, N rounds of { every thread writes a word, __threadfence, one device-scope atomic
//            add per workgroup on a counter, spin until it reaches W x round, read a neighbour workgroup's word }
//   graph:   the same N rounds as N kernel nodes of W workgroups in a captured graph (each node: write a word, read the
//            neighbour's word of the previous node)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) coop(unsigned* counter, unsigned* words, int rounds, unsigned base, unsigned* sink) {
    const unsigned W = gridDim.x;
    unsigned acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        words[blockIdx.x * 256 + threadIdx.x] = r + threadIdx.x;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < base + W * (unsigned)r) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        acc += __hip_atomic_load(&words[((blockIdx.x + 97) % W) * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

__global__ void __launch_bounds__(256) node(unsigned* words, int r, unsigned* sink) {
    const unsigned W = gridDim.x;
    const unsigned v = words[((blockIdx.x + 97) % W) * 256 + threadIdx.x + (r & 1) * W * 256];
    words[blockIdx.x * 256 + threadIdx.x + ((r + 1) & 1) * W * 256] = v + r;
    if (v == 0xffffffffu) sink[0] = v;
}

int main() {
    unsigned *counter, *words, *sink;
    (void)hipMalloc(&counter, 4); (void)hipMalloc(&words, 2 * 1024 * 256 * 4); (void)hipMalloc(&sink, 4);
    (void)hipMemset(counter, 0, 4); (void)hipMemset(words, 0, 2 * 1024 * 256 * 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int rounds = 200;
    unsigned base = 0;
    for (int W : {32, 64, 128, 256}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(coop, dim3(W), dim3(256), 0, st, counter, words, rounds, base, sink);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            base += (unsigned)W * rounds;
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // the same as graph nodes
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(node, dim3(W), dim3(256), 0, st, words, r, sink);
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        float gbest = 1e9f, ebest = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, st);
            (void)hipGraphLaunch(ge, st);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < gbest) gbest = ms;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(node, dim3(W), dim3(256), 0, st, words, r, sink);
            (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < ebest) ebest = ms;
        }
        printf("%3d workgroups: grid barrier %.2f us per round; kernel boundary in a replayed graph %.2f us per node, eager launches %.2f us (%s)\n",
               W, 1e3f * best / rounds, 1e3f * gbest / rounds, 1e3f * ebest / rounds, hipGetErrorString(hipGetLastError()));
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
