// FiLM projections of ALL ResnetBlocks in one launch (gfx950).
//
// Every ResnetBlock of the U-Net maps the same conditioning vector c (B x T, T = 32..96) through its own
// nn.Linear(T, 2 C) to a (scale | shift) pair (reference ddpm.py:184,191-192).  As separate ops that is, per block,
// an addmm, two slice copies, and in the backward two GEMMs, a bias reduction, two zero fills + two copies for the
// chunk's gradient and an accumulation into dc: ~150 launches of 4-7 us per training step for 0.4 MFLOP each.
// Here: one forward launch over the rows of all layers, two backward launches.  fp32 throughout (the vector and the
// projections are never bf16).
//
//   forward    out_i[k][b][ch] = bias_i[k C_i + ch] + sum_t w_i[k C_i + ch][t] c[b][t]        k = 0 scale, 1 shift
//   backward   dw_i[r][t] = sum_b g_i[r, b] c[b][t],  db_i[r] = sum_b g_i[r, b],  dc[b][t] = sum_i sum_r g_i[r, b] w_i[r][t]
//              with g_i[r, b] = g_i[k][b][ch].  dc: per-workgroup partial sums in the workspace, summed in a fixed order
//              by the finish kernel (deterministic, no atomics).
#include "tdx_common.h"

#define FILM_ROWS 64          // rows of the stacked projection matrix per workgroup
#define FILM_MAXB 16          // batch entries per pass

struct FilmFwdTable {
    const float* w[TDX_FILM_MAX_LAYERS];
    const float* b[TDX_FILM_MAX_LAYERS];
    float* out[TDX_FILM_MAX_LAYERS];
    int first[TDX_FILM_MAX_LAYERS + 1];  // first stacked row of layer i; first[n] = total rows
    int n;
};

struct FilmBwdTable {
    const float* w[TDX_FILM_MAX_LAYERS];
    const float* g[TDX_FILM_MAX_LAYERS];
    float* dw[TDX_FILM_MAX_LAYERS];
    float* db[TDX_FILM_MAX_LAYERS];
    int first[TDX_FILM_MAX_LAYERS + 1];
    int n;
};

__device__ __forceinline__ int film_layer_of(const int* first, int n, int row) {
    int i = 0;
    while (i + 1 < n && row >= first[i + 1]) ++i;
    return i;
}

// one thread per stacked row; c staged in LDS
__global__ void __launch_bounds__(FILM_ROWS)
film_fwd_kernel(FilmFwdTable tab, const float* __restrict__ c, int B, int T) {
    extern __shared__ float sc[];  // [B][T]
    for (int i = threadIdx.x; i < B * T; i += FILM_ROWS) sc[i] = c[i];
    __syncthreads();
    const int row = blockIdx.x * FILM_ROWS + threadIdx.x;
    if (row >= tab.first[tab.n]) return;
    const int li = film_layer_of(tab.first, tab.n, row);
    const int r = row - tab.first[li], C = (tab.first[li + 1] - tab.first[li]) >> 1;
    const int k = r / C, ch = r - k * C;
    const float* w = tab.w[li] + (int64_t)r * T;
    const float bias = tab.b[li] ? tab.b[li][r] : 0.f;
    float* out = tab.out[li] + (int64_t)k * B * C + ch;
    for (int b0 = 0; b0 < B; b0 += FILM_MAXB) {
        const int nb = min(FILM_MAXB, B - b0);
        float acc[FILM_MAXB];
#pragma unroll
        for (int b = 0; b < FILM_MAXB; ++b) acc[b] = bias;
        for (int t = 0; t < T; ++t) {
            const float wv = w[t];
#pragma unroll
            for (int b = 0; b < FILM_MAXB; ++b)
                if (b < nb) acc[b] = fmaf(wv, sc[(b0 + b) * T + t], acc[b]);
        }
#pragma unroll
        for (int b = 0; b < FILM_MAXB; ++b)
            if (b < nb) out[(int64_t)(b0 + b) * C] = acc[b];
    }
}

// workgroup = FILM_ROWS stacked rows: dw / db rows, and this row block's share of dc into partial[block][B][T]
__global__ void __launch_bounds__(FILM_ROWS)
film_bwd_kernel(FilmBwdTable tab, const float* __restrict__ c, float* __restrict__ partial, int B, int T) {
    extern __shared__ float sm[];
    float* sc = sm;                       // [B][T]
    float* sg = sm + B * T;               // [FILM_ROWS][B + 1]
    float* sw = sg + FILM_ROWS * (B + 1); // [FILM_ROWS][T + 1]
    for (int i = threadIdx.x; i < B * T; i += FILM_ROWS) sc[i] = c[i];
    const int row = blockIdx.x * FILM_ROWS + threadIdx.x;
    const bool ok = row < tab.first[tab.n];
    int li = 0, r = 0, C = 1;
    if (ok) {
        li = film_layer_of(tab.first, tab.n, row);
        r = row - tab.first[li];
        C = (tab.first[li + 1] - tab.first[li]) >> 1;
    }
    const int k = r / C, ch = r - k * C;
    const float* g = tab.g[li] + (int64_t)k * B * C + ch;
    // loads go out in groups (rolled one-load trips are chains of memory round trips: this kernel is all latency)
    float gsum = 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (ok && b0 + u < B) ? g[(int64_t)min(b0 + u, B - 1) * C] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + u < B) {
                sg[threadIdx.x * (B + 1) + b0 + u] = v[u];
                gsum += v[u];
            }
    }
    const float* w = tab.w[li] + (int64_t)r * T;
    for (int t0 = 0; t0 < T; t0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (ok && t0 + u < T) ? w[min(t0 + u, T - 1)] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (t0 + u < T) sw[threadIdx.x * (T + 1) + t0 + u] = v[u];
    }
    __syncthreads();
    if (ok) {
        float* dw = tab.dw[li] + (int64_t)r * T;
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
            for (int b = 0; b < B; ++b) a = fmaf(sg[threadIdx.x * (B + 1) + b], sc[b * T + t], a);
            dw[t] = a;
        }
        if (tab.db[li]) tab.db[li][r] = gsum;
    }
    // dc share of this row block: thread -> (b, t) pairs
    float* p = partial + (int64_t)blockIdx.x * B * T;
    for (int i = threadIdx.x; i < B * T; i += FILM_ROWS) {
        const int b = i / T, t = i - b * T;
        float a = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < FILM_ROWS; ++rr) a = fmaf(sg[rr * (B + 1) + b], sw[rr * (T + 1) + t], a);
        p[i] = a;
    }
}

__global__ void film_bwd_finish_kernel(const float* __restrict__ partial, float* __restrict__ dc, int nblocks, int BT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BT) return;
    // eight partials' loads in flight per trip, added in the same order as before (a rolled load -> add loop over the
    // ~80 blocks was 80 memory round trips in a row: 21 us for 192 outputs)
    float a = 0.f;
    for (int k0 = 0; k0 < nblocks; k0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partial[(int64_t)min(k0 + u, nblocks - 1) * BT + i];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nblocks) a += t[u];
    }
    dc[i] = a;
}

static int film_rows(const int* channels, int n, int* first) {
    first[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (channels[i] <= 0) return -1;
        first[i + 1] = first[i] + 2 * channels[i];
    }
    return first[n];
}

#define FILM_MAX_LDS (160 * 1024)  // the conditioning vectors of the whole batch sit in LDS (a CU's full 160 KiB at most)
static inline size_t film_bwd_lds(int B, int T) {
    return ((size_t)B * T + (size_t)FILM_ROWS * (B + 1) + (size_t)FILM_ROWS * (T + 1)) * sizeof(float);
}
// 1 if tdx_film_fwd AND tdx_film_bwd take a batch of B conditioning vectors of T features (else project per block)
extern "C" int tdx_film_supported(int B, int T) {
    return B > 0 && T > 0 && (size_t)B * T * sizeof(float) <= FILM_MAX_LDS && film_bwd_lds(B, T) <= FILM_MAX_LDS;
}

extern "C" int tdx_film_fwd(const float* c, int B, int T, const TdxFilmLayer* layers, int n, void* stream) {
    TDX_CHECK_ARG(c && layers && B > 0 && T > 0 && n > 0 && n <= TDX_FILM_MAX_LAYERS);
    FilmFwdTable tab;
    int channels[TDX_FILM_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
        TDX_CHECK_ARG(layers[i].weight && layers[i].out);
        tab.w[i] = layers[i].weight; tab.b[i] = layers[i].bias; tab.out[i] = layers[i].out;
        channels[i] = layers[i].channels;
    }
    tab.n = n;
    const int rows = film_rows(channels, n, tab.first);
    if (rows <= 0) return TDX_EINVAL;
    const size_t lds = (size_t)B * T * sizeof(float);
    if (lds > FILM_MAX_LDS) return TDX_ESHAPE;  // hosts check tdx_film_supported first (ops.film_projections falls back)
    if (lds > 48 * 1024) {
        static size_t attr = 0;
        if (lds > attr) {
            hipError_t e = hipFuncSetAttribute((const void*)film_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FILM_MAX_LDS);
            if (e != hipSuccess) return (int)e;
            attr = FILM_MAX_LDS;
        }
    }
    hipLaunchKernelGGL(film_fwd_kernel, dim3(ceil_div(rows, FILM_ROWS)), dim3(FILM_ROWS), lds, as_stream(stream), tab, c, B, T);
    return tdx_launch_status();
}

extern "C" size_t tdx_film_bwd_workspace_bytes(int B, int T, const int* channels, int n) {
    if (!channels || n <= 0 || n > TDX_FILM_MAX_LAYERS) return 0;
    int first[TDX_FILM_MAX_LAYERS + 1];
    const int rows = film_rows(channels, n, first);
    if (rows <= 0) return 0;
    return (size_t)ceil_div(rows, FILM_ROWS) * B * T * sizeof(float);
}

extern "C" int tdx_film_bwd(const float* c, int B, int T, const TdxFilmGrad* layers, int n, float* dc, void* workspace,
                            void* stream) {
    TDX_CHECK_ARG(c && layers && dc && workspace && B > 0 && T > 0 && n > 0 && n <= TDX_FILM_MAX_LAYERS);
    FilmBwdTable tab;
    int channels[TDX_FILM_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
        TDX_CHECK_ARG(layers[i].weight && layers[i].grad_out && layers[i].grad_weight);
        tab.w[i] = layers[i].weight; tab.g[i] = layers[i].grad_out; tab.dw[i] = layers[i].grad_weight;
        tab.db[i] = layers[i].grad_bias;
        channels[i] = layers[i].channels;
    }
    tab.n = n;
    const int rows = film_rows(channels, n, tab.first);
    if (rows <= 0) return TDX_EINVAL;
    const size_t lds = film_bwd_lds(B, T);
    if (lds > FILM_MAX_LDS) return TDX_ESHAPE;
    if (lds > 48 * 1024) {
        static size_t attr = 0;
        if (lds > attr) {
            hipError_t e = hipFuncSetAttribute((const void*)film_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FILM_MAX_LDS);
            if (e != hipSuccess) return (int)e;
            attr = FILM_MAX_LDS;
        }
    }
    const int nblocks = ceil_div(rows, FILM_ROWS);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(film_bwd_kernel, dim3(nblocks), dim3(FILM_ROWS), lds, st, tab, c, (float*)workspace, B, T);
    int rc = tdx_launch_status();
    if (rc != TDX_OK) return rc;
    hipLaunchKernelGGL(film_bwd_finish_kernel, dim3(ceil_div(B * T, 256)), dim3(256), 0, st, (const float*)workspace, dc, nblocks,
                       B * T);
    return tdx_launch_status();
}
