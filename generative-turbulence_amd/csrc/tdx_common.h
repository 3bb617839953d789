// Shared device helpers for the tdx kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/tdx.h"

typedef __hip_bfloat16 bf16;
typedef _Float16 f16;  // fp16 tensors (TDX_F16): IEEE half storage + fp16 MFMA operands, fp32 accumulation / statistics

#define TDX_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return TDX_EINVAL; \
    } while (0)

static inline int tdx_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TDX_OK : (int)e;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// ---- scalar element access ---------------------------------------------------------
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = __float2bfloat16(v); }
__device__ __forceinline__ float ldf(const f16* p) { return (float)*p; }
__device__ __forceinline__ void stf(f16* p, float v) { *p = (f16)v; }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    bf16 h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&h);
}

// two floats -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const unsigned*>(&h);
}

// two floats -> packed fp16 pair, round to nearest even (v_cvt_f16_f32 x2 + v_pack_b32_f16)
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    const f32x2_t v = {lo, hi};
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    return *reinterpret_cast<const unsigned*>(&h);
}

// ---- 8-element vector access (16 B for bf16, 32 B for f32); pointer must be aligned ----
template <typename T>
struct Vec8;
template <>
struct Vec8<float> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4* q = reinterpret_cast<const float4*>(p);
        float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        float4* q = reinterpret_cast<float4*>(p);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <>
struct Vec8<bf16> {
    float v[8];
    __device__ __forceinline__ void load(const bf16* p) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bf16* p) const {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

template <>
struct Vec8<f16> {
    float v[8];
    __device__ __forceinline__ void load(const f16* p) {
        typedef __attribute__((ext_vector_type(8))) _Float16 h8;
        const h8 u = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)u[i];
    }
    __device__ __forceinline__ void store(f16* p) const {
        typedef __attribute__((ext_vector_type(8))) _Float16 h8;
        h8 u;
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = (_Float16)v[i];
        *reinterpret_cast<h8*>(p) = u;
    }
};

// raw (unconverted) 8-element loads: lets a kernel issue several loads back to back and convert later
template <typename T>
struct Raw8;
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = reinterpret_cast<const float4*>(p)[0];
        b = reinterpret_cast<const float4*>(p)[1];
    }
    __device__ __forceinline__ Vec8<float> get() const {
        Vec8<float> r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
};
template <>
struct Raw8<bf16> {
    uint4 u;
    __device__ __forceinline__ void load(const bf16* p) {
#ifdef TDX_NT_LOADS
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        u = make_uint4(t.x, t.y, t.z, t.w);
#else
        u = *reinterpret_cast<const uint4*>(p);
#endif
    }
    __device__ __forceinline__ Vec8<bf16> get() const {
        Vec8<bf16> r;
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r.v[2 * i] = __uint_as_float(w[i] << 16);
            r.v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
        return r;
    }
};

template <>
struct Raw8<f16> {
    uint4 u;
    __device__ __forceinline__ void load(const f16* p) { u = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ Vec8<f16> get() const {
        typedef __attribute__((ext_vector_type(8))) _Float16 h8;
        const h8 h = __builtin_bit_cast(h8, u);
        Vec8<f16> r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = (float)h[i];
        return r;
    }
};

// ---- the two 16-bit formats of the matrix-core kernels.  Those kernels move operands as raw 16-bit words (LDS-DMA, 16-B
// register pieces, `bf16x8` fragments); what depends on the format is the MFMA opcode, the rounding of fp32 results to 16
// bits, the widening of stored words, and the constant 1.0 of the bias-gradient slots.  HF = false: bfloat16 (TDX_BF16),
// HF = true: IEEE half (TDX_F16; same matrix-pipe cycles on gfx950, 11 instead of 8 significand bits -- the arithmetic of
// the reference's TF32 GPU runs, train.py:144-156).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool HF>
struct H16;
template <>
struct H16<false> {
    typedef bf16 T;
    static constexpr int code = TDX_BF16;
    static __device__ __forceinline__ f32x16_t mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
    static __device__ __forceinline__ bf16x8_t ones() {
        const uint4 u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        return __builtin_bit_cast(bf16x8_t, u);
    }
};
template <>
struct H16<true> {
    typedef f16 T;
    static constexpr int code = TDX_F16;
    static __device__ __forceinline__ f32x16_t mfma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ float lo(unsigned w) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, w)[0];
    }
    static __device__ __forceinline__ float hi(unsigned w) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, w)[1];
    }
    static __device__ __forceinline__ bf16x8_t ones() {
        const uint4 u = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        return __builtin_bit_cast(bf16x8_t, u);
    }
};
static inline bool tdx_is_h16(int dtype) { return dtype == TDX_BF16 || dtype == TDX_F16; }

// ---- wave / block reductions (wave = 64 lanes) -----------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sigmoid / SiLU on the hardware exp2 and reciprocal (1 ulp each): 5 VALU instructions per SiLU
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d silu(x)/dx = s (1 + x (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
    const float s = sigmoid_f(x);
    return s * __builtin_fmaf(x, 1.0f - s, 1.0f);
}

// Workgroups are dealt round-robin to the 8 XCDs of the chip, each with its own L2.  xcd_contiguous maps
// the hardware block id to a logical id such that every XCD owns ONE contiguous range of logical ids
// (a bijection on [0, n)), so that neighbouring tiles -- which share halo voxels -- share an L2.
__device__ __forceinline__ int xcd_contiguous(int bid, int n) {
    const int xcd = bid & 7, k = bid >> 3;
    const int base = n >> 3, rem = n & 7;
    return xcd * base + min(xcd, rem) + k;
}

// Zero fills as kernel launches (tdx_ordered.hip): hipMemsetAsync becomes a memset NODE in a captured graph, which this runtime
// does not order against earlier kernel nodes writing the memory's previous owner -- never use it on a path that can be captured
int tdx_zero_async(void* p, size_t bytes, hipStream_t st);
int tdx_zero2d_async(void* p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t st);

// dtype dispatch for launchers: calls f.template operator()<T>()
#define TDX_DISPATCH_DTYPE(dtype, ...)                 \
    do {                                               \
        if ((dtype) == TDX_F32) {                      \
            typedef float T;                           \
            __VA_ARGS__;                               \
        } else if ((dtype) == TDX_BF16) {              \
            typedef bf16 T;                            \
            __VA_ARGS__;                               \
        } else if ((dtype) == TDX_F16) {               \
            typedef f16 T;                             \
            __VA_ARGS__;                               \
        } else                                         \
            return TDX_EDTYPE;                         \
    } while (0)

// 16-bit format dispatch for the matrix-core launchers: runs the statement with `constexpr bool HF`
#define TDX_DISPATCH_H16(f16, ...)          \
    do {                                    \
        if (f16) {                          \
            constexpr bool HF = true;       \
            __VA_ARGS__;                    \
        } else {                            \
            constexpr bool HF = false;      \
            __VA_ARGS__;                    \
        }                                   \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
