// Fused tail of a training step: global gradient-norm clipping (the reference trains with
// gradient_clip_val = 0.1, config/train.yaml:30-31) + the RAdam update (diffusion.py:210-218,
// torch.optim.RAdam defaults) over ALL parameter tensors in three launches, instead of ~40
// multi-tensor launches + ~70 fills of the stock foreach implementation.
//
// The tensors stay separate allocations (the state_dict keeps the reference's keys); the kernels
// walk a device table of {param, grad, exp_avg, exp_avg_sq} pointers.  Work is cut into chunks of
// OPT_CHUNK elements; chunk c belongs to tensor chunk_tensor[c] at element offset chunk_off[c]
// (host-built once, the shapes never change).  HBM-bound: 1 read pass for the norm, 4 reads +
// 3 writes for the update (fp32).
#include "tdx_common.h"

#define OPT_CHUNK 16384
#define OPT_THREADS 256


__global__ void __launch_bounds__(OPT_THREADS)
opt_sumsq_kernel(const TdxOptTensor* __restrict__ table, const int* __restrict__ chunk_tensor,
                 const int64_t* __restrict__ chunk_off, float* __restrict__ partial) {
    const int c = blockIdx.x;
    const TdxOptTensor t = table[chunk_tensor[c]];
    const float* g = (const float*)t.grad;
    float s = 0.f;
    if (g != nullptr) {
        const int64_t o = chunk_off[c];
        const int64_t n = min((int64_t)OPT_CHUNK, t.numel - o);
        const float* gp = g + o;
        // gradients may be views into a flat all-reduce bucket: only 4-B alignment is guaranteed
        const int64_t n4 = (reinterpret_cast<uintptr_t>(gp) & 15) == 0 ? (n >> 2) : 0;
        for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
            const float4 a = reinterpret_cast<const float4*>(gp)[i];
            s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_THREADS) s += gp[i] * gp[i];
    }
    __shared__ float red[OPT_THREADS / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[c] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0).
// SCALED (loss-scaled gradients, fp16 training): the stored gradients are inv_scale^-1 times the true ones -- out[0] is the
// TRUE norm, out[1] the factor that takes a stored gradient to the clipped true one (inv_scale x clip coefficient), and
// out[2] = 1 when the norm is not finite (an overflow somewhere in the fp16 backward: the step must be skipped), else 0.
template <bool SCALED>
__global__ void __launch_bounds__(OPT_THREADS)
opt_norm_finalize_kernel(const float* __restrict__ partial, int nchunks, float max_norm, float inv_scale, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nchunks; i += OPT_THREADS) s += (double)partial[i];
    __shared__ double red[OPT_THREADS / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
        if (SCALED) norm *= inv_scale;
        const float coef = max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f;
        out[0] = norm;
        if (SCALED) {
            const bool bad = !(fabsf(norm) <= 3.0e38f);  // inf or nan
            out[1] = bad ? 0.f : coef * inv_scale;
            out[2] = bad ? 1.f : 0.f;
        } else {
            out[1] = coef;
        }
    }
}

struct RAdamArgs {
    float lr, beta1, beta2, eps;
    float inv_bc1;   // 1 / (1 - beta1^t)
    float sqrt_bc2;  // sqrt(1 - beta2^t)
    float rect;      // variance rectification term, or < 0 while rho_t <= 5 (plain momentum step)
};

__device__ __forceinline__ void radam_elem(float& p, float g, float& m, float& v, const RAdamArgs& a) {
    m = m + (g - m) * (1.0f - a.beta1);             // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float mhat = m * a.inv_bc1;
    float upd = mhat * a.lr;
    if (a.rect >= 0.f) upd = upd * (a.sqrt_bc2 / (sqrtf(v) + a.eps)) * a.rect;
    p -= upd;
}

template <bool WRITE_GRAD>
__global__ void __launch_bounds__(OPT_THREADS)
opt_radam_kernel(const TdxOptTensor* __restrict__ table, const int* __restrict__ chunk_tensor,
                 const int64_t* __restrict__ chunk_off, const float* __restrict__ clip, const float* __restrict__ skip,
                 RAdamArgs a) {
    const int c = blockIdx.x;
    const TdxOptTensor t = table[chunk_tensor[c]];
    if (t.grad == nullptr) return;  // parameter without a gradient this step: untouched, as torch does
    if (skip != nullptr && *skip != 0.f) return;  // non-finite loss-scaled gradients: nothing is touched (GradScaler's skipped step)
    const float coef = clip ? clip[1] : 1.0f;
    const int64_t o = chunk_off[c];
    const int64_t n = min((int64_t)OPT_CHUNK, t.numel - o);
    float* p = (float*)t.param + o;
    float* g = (float*)t.grad + o;
    float* m = (float*)t.exp_avg + o;
    float* v = (float*)t.exp_avg_sq + o;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;  // gradients may be views into a flat bucket
    const int64_t n4 = aligned ? (n >> 2) : 0;
    for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        gg.x *= coef; gg.y *= coef; gg.z *= coef; gg.w *= coef;
        radam_elem(pp.x, gg.x, mm.x, vv.x, a);
        radam_elem(pp.y, gg.y, mm.y, vv.y, a);
        radam_elem(pp.z, gg.z, mm.z, vv.z, a);
        radam_elem(pp.w, gg.w, mm.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (WRITE_GRAD) reinterpret_cast<float4*>(g)[i] = gg;
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_THREADS) {
        float pp = p[i], gg = g[i] * coef, mm = m[i], vv = v[i];
        radam_elem(pp, gg, mm, vv, a);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (WRITE_GRAD) g[i] = gg;
    }
}

extern "C" int64_t tdx_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int tdx_grad_norm(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                             float max_norm, float* partial, float* out, void* stream) {
    TDX_CHECK_ARG(table && chunk_tensor && chunk_off && partial && out && nchunks > 0);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, st, table, chunk_tensor, chunk_off, partial);
    hipLaunchKernelGGL(opt_norm_finalize_kernel<false>, dim3(1), dim3(OPT_THREADS), 0, st, partial, nchunks, max_norm, 1.0f, out);
    return tdx_launch_status();
}

extern "C" int tdx_grad_norm_scaled(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                                    float max_norm, float inv_scale, float* partial, float* out, void* stream) {
    TDX_CHECK_ARG(table && chunk_tensor && chunk_off && partial && out && nchunks > 0 && inv_scale > 0.f);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, st, table, chunk_tensor, chunk_off, partial);
    hipLaunchKernelGGL(opt_norm_finalize_kernel<true>, dim3(1), dim3(OPT_THREADS), 0, st, partial, nchunks, max_norm, inv_scale, out);
    return tdx_launch_status();
}

static int radam_step_impl(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                           const float* clip, const float* skip, int64_t step, float lr, float beta1, float beta2, float eps,
                           int write_grad, void* stream) {
    TDX_CHECK_ARG(table && chunk_tensor && chunk_off && nchunks > 0 && step >= 1);
    TDX_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f);
    // scalar schedule of torch.optim.radam._single_tensor_radam, in double like the Python floats there
    const double b1 = beta1, b2 = beta2, t = (double)step;
    const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
    const double rho_t = rho_inf - 2.0 * t * pow(b2, t) / bc2;
    RAdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.inv_bc1 = (float)(1.0 / bc1);
    a.sqrt_bc2 = (float)sqrt(bc2);
    a.rect = rho_t > 5.0 ? (float)sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) : -1.0f;
    hipStream_t st = as_stream(stream);
    if (write_grad)
        hipLaunchKernelGGL(opt_radam_kernel<true>, dim3(nchunks), dim3(OPT_THREADS), 0, st, table, chunk_tensor, chunk_off, clip, skip, a);
    else
        hipLaunchKernelGGL(opt_radam_kernel<false>, dim3(nchunks), dim3(OPT_THREADS), 0, st, table, chunk_tensor, chunk_off, clip, skip, a);
    return tdx_launch_status();
}

extern "C" int tdx_radam_step(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                              const float* clip, int64_t step, float lr, float beta1, float beta2, float eps,
                              int write_grad, void* stream) {
    return radam_step_impl(table, chunk_tensor, chunk_off, nchunks, clip, nullptr, step, lr, beta1, beta2, eps, write_grad, stream);
}

extern "C" int tdx_radam_step_scaled(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                                     const float* scaled_norm, int64_t step, float lr, float beta1, float beta2, float eps,
                                     int write_grad, void* stream) {
    TDX_CHECK_ARG(scaled_norm);
    return radam_step_impl(table, chunk_tensor, chunk_off, nchunks, scaled_norm, scaled_norm + 2, step, lr, beta1, beta2, eps,
                           write_grad, stream);
}

// ---- gradient staging of the data-parallel step ---------------------------------------------------------------------------
// The reference's DistributedDataParallel (Lightning strategy, train.py:144-156) copies every gradient into its bucket as
// the autograd hook fires.  Here the hooks only remember (gradient, bucket slice); when a bucket's last gradient has arrived
// ONE launch moves all of them, scaled by 1 / world: 139 five-microsecond launches per step become 7 (0.8 ms of kernel time
// + the gaps between them at B = 6; profiles/r13_ddp_staging.txt).  HBM-bound, 1 read + 1 write of the payload.
// The items travel BY VALUE in the kernel arguments (no table upload, nothing to keep alive; inside a hipGraph capture the
// pointers are part of the node).  Block b serves item i with first[i] <= b < first[i + 1], STAGE_CHUNK elements per block.
#define STAGE_CHUNK 8192
struct StageArgs {
    TdxStageItem item[TDX_STAGE_MAX_ITEMS];
    int first[TDX_STAGE_MAX_ITEMS + 1];
    int n;
    float scale;
};

__global__ void __launch_bounds__(OPT_THREADS) stage_scaled_kernel(const StageArgs a) {
    const int b = blockIdx.x;
    int lo = 0, hi = a.n;  // uniform binary search in the (scalar-register) argument block
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.first[mid] <= b) lo = mid; else hi = mid;
    }
    const int64_t o = (int64_t)(b - a.first[lo]) * STAGE_CHUNK;
    const int64_t n = min((int64_t)STAGE_CHUNK, a.item[lo].n - o);
    const float* __restrict__ s = a.item[lo].src ? (const float*)a.item[lo].src + o : nullptr;
    float* __restrict__ d = (float*)a.item[lo].dst + o;
    const float sc = a.scale;
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s) {
            v = reinterpret_cast<const float4*>(s)[i];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        reinterpret_cast<float4*>(d)[i] = v;
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_THREADS) d[i] = s ? s[i] * sc : 0.f;
}

extern "C" int tdx_stage_scaled(const TdxStageItem* items, int n_items, float scale, void* stream) {
    TDX_CHECK_ARG(items || n_items == 0);
    TDX_CHECK_ARG(n_items >= 0);
    for (int base = 0; base < n_items; base += TDX_STAGE_MAX_ITEMS) {
        StageArgs a;
        a.n = 0;
        a.scale = scale;
        int blocks = 0;
        for (int i = base; i < n_items && a.n < TDX_STAGE_MAX_ITEMS; ++i) {
            TDX_CHECK_ARG(items[i].n >= 0 && (items[i].dst || items[i].n == 0));
            if (items[i].n == 0) continue;
            const int64_t nb = (items[i].n + STAGE_CHUNK - 1) / STAGE_CHUNK;
            TDX_CHECK_ARG(blocks + nb < (int64_t)1 << 30);
            a.item[a.n] = items[i];
            a.first[a.n] = blocks;
            blocks += (int)nb;
            ++a.n;
        }
        if (a.n == 0) continue;
        for (int i = a.n; i <= TDX_STAGE_MAX_ITEMS; ++i) a.first[i] = blocks;
        hipLaunchKernelGGL(stage_scaled_kernel, dim3(blocks), dim3(OPT_THREADS), 0, as_stream(stream), a);
        const int st = tdx_launch_status();
        if (st != 0) return st;
    }
    return 0;
}
