// 1x1x1 convolution / per-row linear layer on NDHWC activations (vector-ALU version).
//   y[r, :] = bias + [x1 | x2][r, :] @ w  (+ add[r, :])
// 64 rows x 64 cols per 256-thread block, 4x4 register tile per thread, K staged through
// LDS in 32-deep slices (x slice stored k-major so the inner loop reads conflict-free
// float4 rows).  f32 accumulation.  These layers are HBM-bound at the U-Net's channel
// counts (4-174 FLOP/B); the MFMA variant lives in tdx_conv3_mfma.hip (taps = 1).
#include "tdx_common.h"
#include "tdx_conv3.h"

#include <stdlib.h>
// MFMA versions (tdx_conv1_mfma.hip); TDX_CONV1_IMPL=direct forces the vector-ALU kernels
bool conv1_mfma_supported(int C1, int C2, int Cout);
int conv1_mfma_fwd_launch(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                          const void* add, void* y, int64_t rows, int Cout, hipStream_t st, const float* gn_stats = nullptr,
                          const float* gn_gamma = nullptr, const float* gn_beta = nullptr, int gn_groups = 1,
                          int64_t gn_voxels = 1, bool hf = false);
bool conv1_wgrad_mfma_supported(int Cin, int Cout);
// max_split / split_stride (TDX_DETERMINISTIC): at most max_split K splits (0 = the launcher's own choice), split k adds into
// dw + k * split_stride (and dbias + k * split_stride): zeroed slabs with ONE contributor per element, summed in order afterwards;
// plan_only: report the number of splits the launch would use (nsplit_out) without launching
int conv1_wgrad_mfma_launch(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                            int64_t rows, bool transposed, hipStream_t st, bool hf = false, int max_split = 0,
                            int64_t split_stride = 0, int* nsplit_out = nullptr, bool plan_only = false);
// fp32 MFMA versions (tdx_conv1_mfma_f32.hip)
bool conv1_mfma_f32_supported(int C1, int C2, int Cout, const float* w, int ldw);
int conv1_mfma_f32_fwd_launch(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                              const void* add, void* y, int64_t rows, int Cout, hipStream_t st);
bool conv1_wgrad_mfma_f32_supported(int Cin, int Cout);
int conv1_wgrad_mfma_f32_launch(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                                int64_t rows, bool transposed, hipStream_t st, int max_split = 0, int64_t split_stride = 0,
                                int* nsplit_out = nullptr, bool plan_only = false);
static bool conv1_force_direct() {
    const char* e = getenv("TDX_CONV1_IMPL");
    return e && e[0] == 'd';
}

#define C1_BM 64
#define C1_BN 64
#define C1_BK 32

template <typename T>
__global__ void __launch_bounds__(256)
conv1_fwd_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2, int C2, const float* __restrict__ w,
                 int ldw, const float* __restrict__ bias, const T* __restrict__ add, T* __restrict__ y, int64_t rows,
                 int Cout) {
    __shared__ float xs[C1_BK][C1_BM + 4];
    __shared__ float ws[C1_BK][C1_BN + 4];
    const int Cin = C1 + C2;
    const int64_t r0 = (int64_t)blockIdx.x * C1_BM;
    const int n0 = blockIdx.y * C1_BN;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < Cin; k0 += C1_BK) {
        // stage x slice: 64 rows x 32 k  (thread: row = tid/4, 8 consecutive k)
        {
            const int rr = tid >> 2, kk = (tid & 3) * 8;
            const int64_t r = r0 + rr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kk + j;
                float v = 0.f;
                if (r < rows && k < Cin) v = (k < C1) ? ldf(x1 + r * C1 + k) : ldf(x2 + r * C2 + (k - C1));
                xs[kk + j][rr] = v;
            }
        }
        // stage w slice: 32 k x 64 n
        {
            const int kk = tid >> 3, nn = (tid & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kk, n = n0 + nn + j;
                ws[kk][nn + j] = (k < Cin && n < Cout) ? w[(size_t)k * ldw + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < C1_BK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&xs[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&ws[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty * 4 + i;
        if (r >= rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= Cout) continue;
            float v = acc[i][j];
            if (bias) v += bias[n];
            if (add) v += ldf(add + r * Cout + n);
            stf(y + r * Cout + n, v);
        }
    }
}

extern "C" int tdx_conv1_fwd(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw,
                             const float* bias, const void* add, void* y, int64_t rows, int Cout, int dtype,
                             void* stream) {
    TDX_CHECK_ARG(x1 && w && y && rows > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && ldw >= Cout);
    TDX_CHECK_ARG(C2 == 0 || x2);
    if (tdx_is_h16(dtype) && !conv1_force_direct() && conv1_mfma_supported(C1, C2, Cout))
        return conv1_mfma_fwd_launch(x1, C1, x2, C2, w, ldw, bias, add, y, rows, Cout, as_stream(stream), nullptr, nullptr, nullptr, 1,
                                     1, dtype == TDX_F16);
    if (dtype == TDX_F32 && !conv1_force_direct() && conv1_mfma_f32_supported(C1, C2, Cout, w, ldw))
        return conv1_mfma_f32_fwd_launch(x1, C1, x2, C2, w, ldw, bias, add, y, rows, Cout, as_stream(stream));
    dim3 grid(ceil_div(rows, C1_BM), ceil_div(Cout, C1_BN));
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv1_fwd_kernel<T>), grid, dim3(256), 0, as_stream(stream),
                                                  (const T*)x1, C1, (const T*)x2, C2, w, ldw, bias, (const T*)add,
                                                  (T*)y, rows, Cout));
    return tdx_launch_status();
}

// y = silu(GroupNorm(h)) + bias + [x1|x2] @ w: the tail of a ResnetBlock with a projected skip in one pass (bf16 / fp16 tensors on
// the matrix-core kernel; TDX_ESHAPE / TDX_EDTYPE otherwise: the caller then runs tdx_conv1_fwd + tdx_gn_apply).
extern "C" int tdx_conv1_fwd_gn(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                                const void* h, const float* stats, const float* gamma, const float* beta, int groups,
                                void* y, int B, int64_t V, int Cout, int dtype, void* stream) {
    TDX_CHECK_ARG(x1 && w && y && h && stats && gamma && beta && B > 0 && V > 0 && C1 > 0 && C2 >= 0 && Cout > 0 && ldw >= Cout);
    TDX_CHECK_ARG((C2 == 0 || x2) && groups > 0 && (Cout % groups) == 0);
    if (!tdx_is_h16(dtype)) return TDX_EDTYPE;
    if (conv1_force_direct() || !conv1_mfma_supported(C1, C2, Cout)) return TDX_ESHAPE;
    return conv1_mfma_fwd_launch(x1, C1, x2, C2, w, ldw, bias, h, y, (int64_t)B * V, Cout, as_stream(stream), stats, gamma,
                                 beta, groups, V, dtype == TDX_F16);
}

// dw[ci][co] = sum_r x[r,ci] dy[r,co]; each block reduces a chunk of rows for one
// 64x64 (ci, co) tile and merges with f32 atomics (dw is zeroed first).
#define C1W_ROWS 4096
template <typename T>
__global__ void __launch_bounds__(256)
conv1_wgrad_kernel(const T* __restrict__ x, int Cin, const T* __restrict__ dy, int Cout, float* __restrict__ dw,
                   int ldw, float* __restrict__ dbias, int64_t rows, int transposed, int64_t rows_per_block,
                   int64_t split_stride) {
    __shared__ float xs[C1_BK][C1_BM + 4];  // [row slice][ci]
    __shared__ float gs[C1_BK][C1_BN + 4];  // [row slice][co]
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t rend = min(rows, rbeg + rows_per_block);
    dw += (int64_t)blockIdx.x * split_stride;  // TDX_DETERMINISTIC: one zeroed slab per row chunk
    if (dbias) dbias += (int64_t)blockIdx.x * split_stride;
    const int ci0 = blockIdx.y * C1_BM, co0 = blockIdx.z * C1_BN;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float bsum = 0.f;  // thread tid < 64 sums column co0 + tid when ci tile == 0

    for (int64_t rs = rbeg; rs < rend; rs += C1_BK) {
        {
            const int rr = tid >> 3, cc = (tid & 7) * 8;
            const int64_t r = rs + rr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = ci0 + cc + j, co = co0 + cc + j;
                xs[rr][cc + j] = (r < rend && ci < Cin) ? ldf(x + r * Cin + ci) : 0.f;
                gs[rr][cc + j] = (r < rend && co < Cout) ? ldf(dy + r * Cout + co) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < C1_BK; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(&xs[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4*>(&gs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        if (dbias && blockIdx.y == 0 && tid < C1_BN) {
#pragma unroll 8
            for (int k = 0; k < C1_BK; ++k) bsum += gs[k][tid];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty * 4 + i;
        if (ci >= Cin) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + tx * 4 + j;
            if (co < Cout) atomicAdd(&dw[transposed ? (size_t)co * ldw + ci : (size_t)ci * ldw + co], acc[i][j]);
        }
    }
    if (dbias && blockIdx.y == 0 && tid < C1_BN && co0 + tid < Cout) atomicAdd(&dbias[co0 + tid], bsum);
}

// One launch of whichever weight-gradient kernel serves (dtype, Cin, Cout).  max_split / split_stride: see above.
static int conv1_wgrad_dispatch(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias, int64_t rows,
                                int dtype, bool transposed, hipStream_t st, int max_split, int64_t split_stride, int* nsplit_out,
                                bool plan_only = false) {
    if (tdx_is_h16(dtype) && !conv1_force_direct() && conv1_wgrad_mfma_supported(Cin, Cout))
        return conv1_wgrad_mfma_launch(x, Cin, dy, Cout, dw, ldw, dbias, rows, transposed, st, dtype == TDX_F16, max_split,
                                       split_stride, nsplit_out, plan_only);
    if (dtype == TDX_F32 && !conv1_force_direct() && conv1_wgrad_mfma_f32_supported(Cin, Cout))
        return conv1_wgrad_mfma_f32_launch(x, Cin, dy, Cout, dw, ldw, dbias, rows, transposed, st, max_split, split_stride,
                                           nsplit_out, plan_only);
    int64_t rpb = C1W_ROWS;
    if (max_split > 0 && ceil_div(rows, rpb) > max_split) rpb = (ceil_div(rows, max_split) + C1_BK - 1) / C1_BK * C1_BK;
    dim3 grid(ceil_div(rows, rpb), ceil_div(Cin, C1_BM), ceil_div(Cout, C1_BN));
    if (nsplit_out) *nsplit_out = (int)grid.x;
    if (plan_only) return TDX_OK;
    TDX_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv1_wgrad_kernel<T>), grid, dim3(256), 0, st, (const T*)x, Cin,
                                                  (const T*)dy, Cout, dw, ldw, dbias, rows, transposed ? 1 : 0, rpb, split_stride));
    return tdx_launch_status();
}

static int conv1_bwd_weight_impl(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                                 int64_t rows, int dtype, bool transposed, bool accumulate, hipStream_t st) {
    const int nrow = transposed ? Cout : Cin, ncol = transposed ? Cin : Cout;
    if (tdx_deterministic()) {
        // K split k adds its tile into slab k of the scratch arena (zeroed; one contributor per element, so the f32 atomics
        // are plain stores in effect), then the slabs are added in order into dw / dbias.  Without an arena: one split.
        const int64_t slab = ((int64_t)nrow * ncol + Cout + 63) / 64 * 64;  // [nrow][ncol] compact, then the bias gradient
        char* arena = (char*)tdx_scratch_ptr();
        const int64_t room = arena ? ((int64_t)tdx_scratch_bytes() - 256) / (int64_t)sizeof(float) / slab : 0;
        int max_split = (int)std::min<int64_t>(room, 256);
        if (max_split >= 2) {
            float* slabs = reinterpret_cast<float*>(arena + 256);
            float* bias_slab = dbias ? slabs + (int64_t)nrow * ncol : nullptr;
            int nsplit = 0;  // what the launcher will use (often far fewer than allowed): only those slabs are zeroed and summed
            int rc = conv1_wgrad_dispatch(x, Cin, dy, Cout, slabs, ncol, bias_slab, rows, dtype, transposed, st, max_split, slab,
                                          &nsplit, true);
            if (rc != TDX_OK) return rc;
            if (nsplit < 1 || nsplit > max_split) return TDX_EINVAL;
            rc = tdx_zero_async(slabs, (size_t)nsplit * slab * sizeof(float), st);
            if (rc != TDX_OK) return rc;
            int launched = 0;
            rc = conv1_wgrad_dispatch(x, Cin, dy, Cout, slabs, ncol, bias_slab, rows, dtype, transposed, st, max_split, slab, &launched);
            if (rc != TDX_OK) return rc;
            if (launched != nsplit) return TDX_EINVAL;
            rc = ordered_sum_launch(slabs, nsplit, slab, dw, nrow, ncol, ldw, accumulate, st);
            if (rc != TDX_OK || !dbias) return rc;
            return ordered_sum_launch(slabs + (int64_t)nrow * ncol, nsplit, slab, dbias, 1, Cout, Cout, accumulate, st);
        }
    }
    const int one_split = tdx_deterministic() ? 1 : 0;
    if (!accumulate) {
        // zero the block that is written (ldw may exceed the row length for sub-blocks)
        int e = tdx_zero2d_async(dw, (size_t)ldw * sizeof(float), (size_t)ncol * sizeof(float), (size_t)nrow, st);
        if (e != TDX_OK) return e;
        if (dbias) {
            e = tdx_zero_async(dbias, (size_t)Cout * sizeof(float), st);
            if (e != TDX_OK) return e;
        }
    }
    return conv1_wgrad_dispatch(x, Cin, dy, Cout, dw, ldw, dbias, rows, dtype, transposed, st, one_split, 0, nullptr);
}

extern "C" int tdx_conv1_bwd_weight(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw,
                                    float* dbias, int64_t rows, int dtype, void* stream) {
    TDX_CHECK_ARG(x && dy && dw && rows > 0 && Cin > 0 && Cout > 0 && ldw >= Cout);
    return conv1_bwd_weight_impl(x, Cin, dy, Cout, dw, ldw, dbias, rows, dtype, false, false, as_stream(stream));
}

extern "C" int tdx_conv1_bwd_weight_oc(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw,
                                       float* dbias, int accumulate, int64_t rows, int dtype, void* stream) {
    TDX_CHECK_ARG(x && dy && dw && rows > 0 && Cin > 0 && Cout > 0 && ldw >= Cin);
    return conv1_bwd_weight_impl(x, Cin, dy, Cout, dw, ldw, dbias, rows, dtype, true, accumulate != 0, as_stream(stream));
}
