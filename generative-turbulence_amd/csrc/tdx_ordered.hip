// TDX_DETERMINISTIC=1: run-to-run reproducible parameter gradients.
//
// By default the cross-workgroup merges of the small parameter gradients are fp32 atomics (bias gradients of the 3x3x3 convs,
// weight / bias gradients of the 1x1 convs, of the encoders / decoder and of the first conv): their sums depend on arrival
// order in the last bits (4e-7 relative, profiles/r13_determinism_probe.txt).  With the switch every such merge goes through
// per-split partials that are added in a FIXED order by the two kernels below, and the halo shell takes its ordered route
// (TDX_SHELL_DETERMINISTIC): a training step then produces the same bits every run.  The reference sets no determinism flag
// of its own; under torch the counterpart would be torch.use_deterministic_algorithms.
//
// The f64 atomics of the forward go as well: tdx_conv3_fwd_gn takes the conv and then the statistics pass over its result
// (no moments from the conv epilogues), whose blocks store their own f64 tables for an ordered merge (tdx_groupnorm.hip);
// the loss adds block partials rounded to a 2^-20 grid, on which f64 additions are exact -- order-independent -- up to 2^33.
// tdx_conv3_fwd_partial (the sampler's first conv) does the same: a sampling run is bit-reproducible too.
#include "tdx_common.h"
#include "tdx_conv3.h"

bool tdx_deterministic() {
    const char* e = getenv("TDX_DETERMINISTIC");  // read per call: a test / training-run switch
    return e && atoi(e) != 0;
}

// dst[r * ld + c] (+)= sum_k slabs[k * stride + r * cols + c], k ascending.  One element per 32 lanes when there are many
// slabs (lane l adds slabs l, l + 32, ... in order, then a fixed butterfly), else one element per thread.
template <int LPE>
__global__ void __launch_bounds__(256) ordered_sum_kernel(const float* __restrict__ slabs, int nslab, int64_t stride,
                                                         float* __restrict__ dst, int64_t n, int cols, int64_t ld, int add) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e = gid / LPE;
    const int l = (int)(gid % LPE);
    float v = 0.f;
    if (e < n) {
        const float* p = slabs + e;
        int k = l;
        for (; k + 3 * LPE < nslab; k += 4 * LPE) {  // four loads in flight, added in slab order
            const float a = p[(int64_t)k * stride], b = p[(int64_t)(k + LPE) * stride], c = p[(int64_t)(k + 2 * LPE) * stride],
                        d = p[(int64_t)(k + 3 * LPE) * stride];
            v = (((v + a) + b) + c) + d;
        }
        for (; k < nslab; k += LPE) v += p[(int64_t)k * stride];
    }
    if (LPE > 1) {
#pragma unroll
        for (int o = LPE / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);  // same tree every run; all lanes end with the sum
    }
    if (e < n && l == 0) {
        float* d = dst + (e / cols) * ld + (e % cols);
        *d = add ? *d + v : v;
    }
}

int ordered_sum_launch(const float* slabs, int nslab, int64_t stride, float* dst, int rows, int cols, int64_t ld, bool add,
                       hipStream_t st) {
    const int64_t n = (int64_t)rows * cols;
    if (n <= 0 || nslab <= 0) return TDX_OK;
    if (nslab > 32)
        hipLaunchKernelGGL(ordered_sum_kernel<32>, dim3((unsigned)ceil_div(n * 32, (int64_t)256)), dim3(256), 0, st, slabs, nslab, stride,
                           dst, n, cols, ld, add ? 1 : 0);
    else
        hipLaunchKernelGGL(ordered_sum_kernel<1>, dim3((unsigned)ceil_div(n, (int64_t)256)), dim3(256), 0, st, slabs, nslab, stride, dst, n,
                           cols, ld, add ? 1 : 0);
    return tdx_launch_status();
}

// Bias gradient of a conv over an NDHWC gradient: dbias[c] = sum_v dy[v][c], in a fixed order.
// Pass 1: block b sums voxels [b * vpb, (b + 1) * vpb) -- a thread walks its voxels (v0 + r, v0 + r + rows, ...) for its 8
// channels, the block's rows are added in row order through LDS -> part[b][C].  Pass 2: ordered_sum over the blocks.
#define OB_THREADS 256
template <typename T>
__global__ void __launch_bounds__(OB_THREADS) bias_partial_kernel(const T* __restrict__ dy, int64_t nvox, int C, int64_t vpb,
                                                                  float* __restrict__ part) {
    __shared__ float red[OB_THREADS][9];
    const int L = C >> 3, rows = OB_THREADS / L;
    const int lc = threadIdx.x % L, r = threadIdx.x / L;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const int64_t v0 = (int64_t)blockIdx.x * vpb, v1 = min(nvox, v0 + vpb);
    if (r < rows)
        for (int64_t v = v0 + r; v < v1; v += rows) {
            Vec8<T> g;
            g.load(dy + v * C + lc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += g.v[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = r < rows ? s[j] : 0.f;
    __syncthreads();
    for (int o = threadIdx.x; o < C; o += OB_THREADS) {
        const int l = o >> 3, j = o & 7;
        float t = 0.f;
        for (int q = 0; q < rows; ++q) t += red[q * L + l][j];
        part[(int64_t)blockIdx.x * C + o] = t;
    }
}

size_t bias_grad_ordered_scratch_floats(int C) { return (size_t)256 * C; }

int bias_grad_ordered_launch(const void* dy, int64_t nvox, int C, int dtype, float* dbias, float* part, size_t part_floats,
                             hipStream_t st) {
    if ((C % 8) || C > 8 * OB_THREADS || part_floats < (size_t)C) return TDX_ESHAPE;
    int64_t nblk = std::min<int64_t>(256, (int64_t)(part_floats / C));
    int64_t vpb = ceil_div(nvox, nblk);
    vpb = vpb < 64 ? 64 : vpb;
    nblk = ceil_div(nvox, vpb);
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bias_partial_kernel<T>), dim3((unsigned)nblk), dim3(OB_THREADS), 0, st, (const T*)dy,
                                                  nvox, C, vpb, part));
    int rc = tdx_launch_status();
    if (rc != TDX_OK) return rc;
    return ordered_sum_launch(part, (int)nblk, C, dbias, 1, C, C, false, st);
}

// ---- zero fill as a KERNEL ------------------------------------------------------------------------------------------------
// hipMemsetAsync inside a captured hipGraph becomes a memset node, and on this runtime a small memset node is not ordered
// against EARLIER kernel nodes that still write the previous owner of the same (graph-pool) memory: replayed, the fill can land
// before such a late write, which then sits in the freshly "zeroed" accumulator -- one garbage element in the encoder / decoder
// gradients from the second replay of a captured training step on, found when the captured step stopped forking its weight
// gradients onto a side stream (which had hidden it: record_stream kept those blocks from being reused).  torch itself zeroes
// with fill kernels.  Every zero fill of the library goes through these two launches; a kernel node is ordered like any other.
__global__ void __launch_bounds__(256) zero_bytes_kernel(unsigned char* __restrict__ p, size_t bytes) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i >= bytes) return;
    if (i + 16 <= bytes && (reinterpret_cast<uintptr_t>(p + i) & 15) == 0) {
        *reinterpret_cast<uint4*>(p + i) = make_uint4(0, 0, 0, 0);
    } else {
        const size_t e = i + 16 < bytes ? i + 16 : bytes;
        for (size_t k = i; k < e; ++k) p[k] = 0;
    }
}
__global__ void __launch_bounds__(256) zero_rows_kernel(unsigned char* __restrict__ p, size_t pitch, size_t width, size_t rows) {
    const size_t per_row = (width + 3) / 4;  // 4-byte pieces (width and pitch are multiples of 4: fp32 matrices)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_row * rows) return;
    const size_t r = i / per_row, c = (i - r * per_row) * 4;
    *reinterpret_cast<unsigned*>(p + r * pitch + c) = 0u;
}
int tdx_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return TDX_OK;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, st, (unsigned char*)p, bytes);
    return tdx_launch_status();
}
int tdx_zero2d_async(void* p, size_t pitch, size_t width, size_t rows, hipStream_t st) {
    if (width == 0 || rows == 0) return TDX_OK;
    if ((pitch % 4) || (width % 4) || (reinterpret_cast<uintptr_t>(p) % 4)) return TDX_EINVAL;
    const size_t n = (width / 4) * rows;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned char*)p, pitch, width, rows);
    return tdx_launch_status();
}
