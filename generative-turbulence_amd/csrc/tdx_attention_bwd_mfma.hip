// bf16 / fp16 MFMA flash attention, backward (long token sequences; BASELINE config 5 sizes).  gfx950.
//
// Same data layout and the same "swapped" MFMA orientation as the forward kernel
// (tdx_attention_mfma.hip): a lane owns ONE row of the side its workgroup keeps resident, so every
// per-row scalar (log-sum-exp, delta) is a per-lane register and the probability tile that comes out of
// one MFMA is, converted to bf16 in registers, directly the B operand of the next.
//
// With P = softmax(Q K^T / sqrt(D)) (recomputed from the saved log-sum-exp), dP = dO V^T,
// dS = P o (dP - delta), delta_i = dO_i . O_i:
//
//   dQ kernel   workgroup = 256 queries (a wave = 2 x 32), walks the keys in staged tiles of 64:
//               S^T  = K Q^T            (Q resident, pre-scaled by log2(e)/sqrt(D))
//               dP^T = V dO^T           (dO resident)
//               dS^T = exp2(S^T - lse) o (dP^T - delta)          per-lane lse, delta
//               dQ^T += K^T dS^T        (K^T fragments: transposed LDS reads of the row-major K tile)
//   dK/dV kernel  workgroup = 256 keys, walks the queries in staged tiles of 64 (Q, dO, lse, delta):
//               S  = Q K^T,  dP = dO V^T   (K pre-scaled and V resident as B operands)
//               P  = exp2(S - lse_row),  dS = P o (dP - delta_row)     per-register lse, delta
//               dV^T += dO^T P,  dK^T += Q^T dS   (dO^T / Q^T fragments: transposed LDS reads)
//
// 6 + 8 MFMAs per 32 x 32 tile against the forward's 4; no atomics, no running maximum.
#include "tdx_common.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define FB_D 32
#define FB_RB 256          // resident rows per workgroup
// TW = resident 32-row tiles per wave: 2 (the default: 4 waves per workgroup, 2 waves per SIMD, ~250 / 230 registers) or 1
// (round 6 experiment: 8 waves per workgroup, 3-4 waves per SIMD at <= 168 / 128 registers, as the forward kernel since round 5).
// The ablations (profiles/r13_attention_bwd_ablation.txt) put 2.4 of the 11.5 ms in vector-ALU work next to a 9.1-ms floor of
// the 14 MFMAs per tile pair; more waves per SIMD do not hide it (10.99-12.3 ms): kept as an A/B knob (TDX_ATTN_BWD_TW).
// The walked side is staged in tiles of FB_T(TW) rows, one 16-B piece per thread.
#define FB_WAVES(TW) (FB_RB / (32 * (TW)))
// waves per SIMD the one-tile-per-wave kernels are compiled for (HIP: the second __launch_bounds__ argument is waves per
// execution unit): 4 = at most 128 registers, 3 = at most 168
#ifndef FB_MINW_DQ
#define FB_MINW_DQ 4
#endif
#ifndef FB_MINW_DKV
#define FB_MINW_DKV 4
#endif
#define FB_T(TW) (16 * FB_WAVES(TW))
#define FB_LOG2E 1.4426950408889634f
// Diagnostic builds only (tools/attn_bwd_ablation.sh compiles this file with -DFB_ABL=<bits>): what bounds the two kernels.
//   1 no exponentials (the score stands in for P)          2 no elementwise product dS = P o dP
//   4 no fp32 -> 16-bit conversions of P / dS (a resident fragment stands in as the B operand)
//   8 no dP products (2 of the 6 / 8 MFMAs per tile)      16 no transposed LDS fragment reads (K^T, Q^T, dO^T)
// The product build defines none of it.
#ifndef FB_ABL
#define FB_ABL 0
#endif


// operand element: bf16 (the model's bf16 mode) or fp16 (BASELINE configs[4]); see tdx_attention_mfma.hip
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct AttnBf16 {
    typedef bf16 T;
    typedef bf16x8 V8;
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct AttnF16 {
    typedef f16 T;
    typedef f16x8 V8;
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack_f16x2(a, b); }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// 64-B rows: chunk c of row r at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 fragments)
__device__ __forceinline__ int fb_sw64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

template <typename V8>
__device__ __forceinline__ V8 fb_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(V8, r);
}

// accumulator registers 8 s .. 8 s + 7 of a 32 x 32 tile -> bf16 B operand of k step s
template <typename E>
__device__ __forceinline__ typename E::V8 fb_as_b(const f32x16& t, int s) {
    return __builtin_bit_cast(typename E::V8, make_uint4(E::pack2(t[8 * s], t[8 * s + 1]), E::pack2(t[8 * s + 2], t[8 * s + 3]),
                                                         E::pack2(t[8 * s + 4], t[8 * s + 5]), E::pack2(t[8 * s + 6], t[8 * s + 7])));
}

// resident-side fragment as a B operand (col = row r of the side, k = d), optionally scaled
template <typename E>
__device__ __forceinline__ typename E::V8 fb_row_frag(const typename E::T* row, float scale) {
    Vec8<typename E::T> v;
    v.load(row);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = E::pack2(v.v[2 * e] * scale, v.v[2 * e + 1] * scale);
    return __builtin_bit_cast(typename E::V8, make_uint4(w[0], w[1], w[2], w[3]));
}

// ------------------------------------------------------------------ dQ ---------------------------
template <typename E, int TW>
__global__ void __launch_bounds__(64 * FB_WAVES(TW), TW == 1 ? FB_MINW_DQ : 2)
attn_bwd_dq_mfma_kernel(const typename E::T* __restrict__ qkv, const typename E::T* __restrict__ dout, const float* __restrict__ lse,
                        const float* __restrict__ delta, typename E::T* __restrict__ dqkv, int N, int H) {
    constexpr int T = FB_T(TW), RW = 32 * TW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * T * 64];
    unsigned char* sK = smem;                  // K tile, swizzled chunks (S^T = K Q^T: A operand rows = keys)
    unsigned char* sKp = smem + T * 64;        // K tile, plain rows (K^T fragments by transposed reads)
    unsigned char* sV = smem + 2 * T * 64;     // V tile, swizzled chunks (dP^T = V dO^T)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int ld = 3 * H * FB_D;
    const typename E::T* base = qkv + (int64_t)b * N * ld;
    const int q0 = blockIdx.x * FB_RB + wave * RW;
    const float sm_scale = rsqrtf((float)FB_D);

    typename E::V8 qf[TW][2], gf[TW][2];  // Q (pre-scaled) and dO as B operands: col = query r, k = d
    // -lse log2(e) and -delta of the lane's query, broadcast over the 16 accumulator registers: as the C operands of the
    // score / dP products they deliver S^T - lse and dP^T - delta straight from the matrix pipe (the kernel is bound by its
    // vector-ALU work: that was 32 subtractions and 32 zeroing moves of ~110 issue slots per 32 x 32 tile)
    f32x16 nl[TW], nd[TW];
    f32x16 dq[TW];               // dQ^T[d][q]
#pragma unroll
    for (int qt = 0; qt < TW; ++qt) {
        const int q = min(q0 + qt * 32 + r, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qt][ks] = fb_row_frag<E>(base + (int64_t)q * ld + h * FB_D + ks * 16 + hh * 8, FB_LOG2E * sm_scale);
            gf[qt][ks] = fb_row_frag<E>(dout + ((int64_t)b * N + q) * (H * FB_D) + h * FB_D + ks * 16 + hh * 8, 1.0f);
        }
        const float ml = -lse[((int64_t)b * H + h) * N + q] * FB_LOG2E, md = -delta[((int64_t)b * H + h) * N + q];
#pragma unroll
        for (int i = 0; i < 16; ++i) { dq[qt][i] = 0.f; nl[qt][i] = ml; nd[qt][i] = md; }
    }

    const int st_row = tid >> 2, st_c = tid & 3;
    uint4 kreg, vreg;
    auto load_tile = [&](int k0) {
        const int key = min(k0 + st_row, N - 1);
        const typename E::T* kp = base + (int64_t)key * ld + H * FB_D + h * FB_D + st_c * 8;
        kreg = *reinterpret_cast<const uint4*>(kp);
        vreg = *reinterpret_cast<const uint4*>(kp + H * FB_D);
    };
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int t_col = (16 * (g & 1) + 4 * tp) * 2;

    load_tile(0);
    for (int k0 = 0; k0 < N; k0 += T) {
        __syncthreads();
        *reinterpret_cast<uint4*>(sK + fb_sw64(st_row, st_c)) = kreg;
        *reinterpret_cast<uint4*>(sKp + st_row * 64 + st_c * 16) = kreg;
        *reinterpret_cast<uint4*>(sV + fb_sw64(st_row, st_c)) = vreg;
        __syncthreads();
        if (k0 + T < N) load_tile(k0 + T);

        auto key_block = [&](int kb, auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            typename E::V8 kf[2], vf[2], ktf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[ks] = *reinterpret_cast<const typename E::V8*>(sK + fb_sw64(kb * 32 + r, 2 * ks + hh));
                vf[ks] = *reinterpret_cast<const typename E::V8*>(sV + fb_sw64(kb * 32 + r, 2 * ks + hh));
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // K^T: rows = d, k = key 16 s + 8 (j >> 2) + 4 hh + (j & 3)
                const unsigned char* kp = sKp + (kb * 32 + 16 * s + 4 * (g >> 1) + tq) * 64 + t_col;
                ktf[s] = (FB_ABL & 16) ? kf[s] : fb_tr_frag<typename E::V8>(kp, kp + 8 * 64);
            }
#pragma unroll
            for (int qt = 0; qt < TW; ++qt) {
                f32x16 st = E::mfma(kf[0], qf[qt][0], nl[qt]);   // S^T - lse
                st = E::mfma(kf[1], qf[qt][1], st);
                f32x16 dp = nd[qt];
                if (!(FB_ABL & 8)) {
                    dp = E::mfma(vf[0], gf[qt][0], nd[qt]);   // dP^T - delta
                    dp = E::mfma(vf[1], gf[qt][1], dp);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float p = (FB_ABL & 1) ? st[i] : __builtin_amdgcn_exp2f(st[i]);
                    if (TAIL && k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) p = 0.f;  // keys beyond N
                    st[i] = (FB_ABL & 2) ? p : p * dp[i];  // dS^T
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) dq[qt] = E::mfma(ktf[s], (FB_ABL & 4) ? qf[qt][s] : fb_as_b<E>(st, s), dq[qt]);
                if (FB_ABL & 4) dq[qt][0] += st[0] + st[15];  // (keeps the elementwise work alive)
            }
        };
        if (k0 + T <= N) {
#pragma unroll
            for (int kb = 0; kb < T / 32; ++kb) key_block(kb, std::false_type{});
        } else {
#pragma unroll
            for (int kb = 0; kb < T / 32; ++kb) key_block(kb, std::true_type{});
        }
    }
    // dQ[q][h*D + d] = scale * dQ^T[d][q];  lane holds d = (i & 3) + 8 (i >> 2) + 4 hh
#pragma unroll
    for (int qt = 0; qt < TW; ++qt) {
        const int q = q0 + qt * 32 + r;
        if (q < N) {
            typename E::T* op = dqkv + ((int64_t)b * N + q) * ld + h * FB_D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 v = make_uint2(E::pack2(dq[qt][4 * j] * sm_scale, dq[qt][4 * j + 1] * sm_scale),
                                           E::pack2(dq[qt][4 * j + 2] * sm_scale, dq[qt][4 * j + 3] * sm_scale));
                *reinterpret_cast<uint2*>(op + 8 * j + 4 * hh) = v;
            }
        }
    }
}

// ------------------------------------------------------------------ dK, dV -----------------------
template <typename E, int TW>
__global__ void __launch_bounds__(64 * FB_WAVES(TW), TW == 1 ? FB_MINW_DKV : 2)
attn_bwd_dkv_mfma_kernel(const typename E::T* __restrict__ qkv, const typename E::T* __restrict__ dout, const float* __restrict__ lse,
                         const float* __restrict__ delta, typename E::T* __restrict__ dqkv, int N, int H) {
    constexpr int T = FB_T(TW), RW = 32 * TW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * T * 64 + 2 * T * 4];
    unsigned char* sQ = smem;                   // Q tile, swizzled (S = Q K^T: A operand rows = queries)
    unsigned char* sQp = smem + T * 64;         // Q tile, plain rows (Q^T fragments)
    unsigned char* sG = smem + 2 * T * 64;      // dO tile, swizzled (dP = dO V^T)
    unsigned char* sGp = smem + 3 * T * 64;     // dO tile, plain rows (dO^T fragments)
    float* sL = reinterpret_cast<float*>(smem + 4 * T * 64);  // -lse * log2(e) of the tile's queries (-inf beyond N)
    float* sD = sL + T;                                        // -delta

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int ld = 3 * H * FB_D;
    const typename E::T* base = qkv + (int64_t)b * N * ld;
    const int j0 = blockIdx.x * FB_RB + wave * RW;
    const float sm_scale = rsqrtf((float)FB_D);

    typename E::V8 kf[TW][2], vf[TW][2];  // K (pre-scaled) and V as B operands: col = key r, k = d
    f32x16 dk[TW], dv[TW];        // dK^T[d][key], dV^T[d][key]
#pragma unroll
    for (int kt = 0; kt < TW; ++kt) {
        const int key = min(j0 + kt * 32 + r, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kt][ks] = fb_row_frag<E>(base + (int64_t)key * ld + H * FB_D + h * FB_D + ks * 16 + hh * 8, FB_LOG2E * sm_scale);
            vf[kt][ks] = fb_row_frag<E>(base + (int64_t)key * ld + 2 * H * FB_D + h * FB_D + ks * 16 + hh * 8, 1.0f);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) dk[kt][i] = dv[kt][i] = 0.f;
    }

    const int st_row = tid >> 2, st_c = tid & 3;
    uint4 qreg, greg;
    float lreg = 0.f, dreg = 0.f;
    auto load_tile = [&](int i0) {
        const int q = min(i0 + st_row, N - 1);
        qreg = *reinterpret_cast<const uint4*>(base + (int64_t)q * ld + h * FB_D + st_c * 8);
        greg = *reinterpret_cast<const uint4*>(dout + ((int64_t)b * N + q) * (H * FB_D) + h * FB_D + st_c * 8);
        if (tid < T) {
            const int qq = i0 + tid;
            lreg = qq < N ? -lse[((int64_t)b * H + h) * N + qq] * FB_LOG2E : -INFINITY;  // exp2(s - inf) = 0: no query there
            dreg = qq < N ? -delta[((int64_t)b * H + h) * N + qq] : 0.f;
        }
    };
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int t_col = (16 * (g & 1) + 4 * tp) * 2;

    load_tile(0);
    for (int i0 = 0; i0 < N; i0 += T) {
        __syncthreads();
        *reinterpret_cast<uint4*>(sQ + fb_sw64(st_row, st_c)) = qreg;
        *reinterpret_cast<uint4*>(sQp + st_row * 64 + st_c * 16) = qreg;
        *reinterpret_cast<uint4*>(sG + fb_sw64(st_row, st_c)) = greg;
        *reinterpret_cast<uint4*>(sGp + st_row * 64 + st_c * 16) = greg;
        if (tid < T) { sL[tid] = lreg; sD[tid] = dreg; }
        __syncthreads();
        if (i0 + T < N) load_tile(i0 + T);

#pragma unroll
        for (int qb = 0; qb < T / 32; ++qb) {
            typename E::V8 qa[2], ga[2], qtf[2], gtf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // A operands: row = query r, k = d
                qa[ks] = *reinterpret_cast<const typename E::V8*>(sQ + fb_sw64(qb * 32 + r, 2 * ks + hh));
                ga[ks] = *reinterpret_cast<const typename E::V8*>(sG + fb_sw64(qb * 32 + r, 2 * ks + hh));
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {  // Q^T, dO^T: rows = d, k = query 16 s + 8 (j >> 2) + 4 hh + (j & 3)
                const int off = (qb * 32 + 16 * s + 4 * (g >> 1) + tq) * 64 + t_col;
                qtf[s] = (FB_ABL & 16) ? qa[s] : fb_tr_frag<typename E::V8>(sQp + off, sQp + off + 8 * 64);
                gtf[s] = (FB_ABL & 16) ? ga[s] : fb_tr_frag<typename E::V8>(sGp + off, sGp + off + 8 * 64);
            }
            // per-register row scalars: register i <-> query row (i & 3) + 8 (i >> 2) + 4 hh of the block.  The sixteen
            // (negated) values ARE the C operands of the score / dP products: S - lse and dP - delta come out of the
            // matrix pipe, the vector ALU keeps the exponential, one product and the conversions
            f32x16 lrow, drow;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 l4 = *reinterpret_cast<const float4*>(sL + qb * 32 + 8 * c + 4 * hh);
                const float4 d4 = *reinterpret_cast<const float4*>(sD + qb * 32 + 8 * c + 4 * hh);
                lrow[4 * c] = l4.x; lrow[4 * c + 1] = l4.y; lrow[4 * c + 2] = l4.z; lrow[4 * c + 3] = l4.w;
                drow[4 * c] = d4.x; drow[4 * c + 1] = d4.y; drow[4 * c + 2] = d4.z; drow[4 * c + 3] = d4.w;
            }
#pragma unroll
            for (int kt = 0; kt < TW; ++kt) {
                f32x16 st = E::mfma(qa[0], kf[kt][0], lrow);
                st = E::mfma(qa[1], kf[kt][1], st);
                f32x16 dp = drow;
                if (!(FB_ABL & 8)) {
                    dp = E::mfma(ga[0], vf[kt][0], drow);
                    dp = E::mfma(ga[1], vf[kt][1], dp);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (!(FB_ABL & 1)) st[i] = __builtin_amdgcn_exp2f(st[i]);  // P
                    if (!(FB_ABL & 2)) dp[i] = st[i] * dp[i];                   // dS
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dv[kt] = E::mfma(gtf[s], (FB_ABL & 4) ? kf[kt][s] : fb_as_b<E>(st, s), dv[kt]);
                    dk[kt] = E::mfma(qtf[s], (FB_ABL & 4) ? vf[kt][s] : fb_as_b<E>(dp, s), dk[kt]);
                }
                if (FB_ABL & 4) dv[kt][0] += st[0] + st[15] + dp[0] + dp[15];  // (keeps the elementwise work alive)
            }
        }
    }
    // dK[key][..] = scale * dK^T, dV[key][..] = dV^T;  lane holds d = (i & 3) + 8 (i >> 2) + 4 hh
#pragma unroll
    for (int kt = 0; kt < TW; ++kt) {
        const int key = j0 + kt * 32 + r;
        if (key < N) {
            typename E::T* okp = dqkv + ((int64_t)b * N + key) * ld + H * FB_D + h * FB_D;
            typename E::T* ovp = okp + H * FB_D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<uint2*>(okp + 8 * j + 4 * hh) =
                    make_uint2(E::pack2(dk[kt][4 * j] * sm_scale, dk[kt][4 * j + 1] * sm_scale),
                               E::pack2(dk[kt][4 * j + 2] * sm_scale, dk[kt][4 * j + 3] * sm_scale));
                *reinterpret_cast<uint2*>(ovp + 8 * j + 4 * hh) =
                    make_uint2(E::pack2(dv[kt][4 * j], dv[kt][4 * j + 1]), E::pack2(dv[kt][4 * j + 2], dv[kt][4 * j + 3]));
            }
        }
    }
}

int attn_bwd_mfma_launch(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int B, int N,
                         int H, int dtype, hipStream_t st) {
    dim3 grid(ceil_div(N, FB_RB), B * H);
    // TDX_ATTN_BWD_TW (A/B knob, read per call): tiles per wave of the dQ / dK-dV kernel, "<dq><dkv>", e.g. 21.  Default 22
    // (two tiles per wave, two waves per SIMD): one tile per wave at three or four waves per SIMD measured 10.99-12.3 ms
    // against 11.4-11.5 (profiles/r13_attention_bwd_ablation.txt) -- more waves do not hide the vector work either
    const char* env = getenv("TDX_ATTN_BWD_TW");
    const int twq = env && env[0] == '1' ? 1 : 2, twk = env && env[0] && env[1] == '1' ? 1 : 2;
#define FB_GO(E, T)                                                                                                            \
    do {                                                                                                                       \
        if (twq == 2) hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<E, 2>), grid, dim3(64 * FB_WAVES(2)), 0, st, (const T*)qkv,   \
                                         (const T*)dout, lse, delta, (T*)dqkv, N, H);                                          \
        else hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<E, 1>), grid, dim3(64 * FB_WAVES(1)), 0, st, (const T*)qkv,            \
                                (const T*)dout, lse, delta, (T*)dqkv, N, H);                                                   \
        if (twk == 2) hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<E, 2>), grid, dim3(64 * FB_WAVES(2)), 0, st, (const T*)qkv,  \
                                         (const T*)dout, lse, delta, (T*)dqkv, N, H);                                          \
        else hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<E, 1>), grid, dim3(64 * FB_WAVES(1)), 0, st, (const T*)qkv,           \
                                (const T*)dout, lse, delta, (T*)dqkv, N, H);                                                   \
    } while (0)
    if (dtype == TDX_F16) FB_GO(AttnF16, f16); else FB_GO(AttnBf16, bf16);
#undef FB_GO
    return tdx_launch_status();
}
