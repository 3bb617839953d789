// bf16 / fp16 MFMA flash attention (forward) for long token sequences (BASELINE config 5:
// N = 96*32*24 = 73 728 voxels, 4 heads x 32).  gfx950.
//
// qkv is the NDHWC to_qkv output [B][N][3*H*D] (q | k | v thirds, head-major), D = 32.
// One wave owns 64 queries (two 32-query tiles) of one (b, h); a workgroup = 4 waves = 256
// queries, and walks the keys in tiles of 64 staged through LDS (K and V, 4 KB each, shared
// by the 4 waves; the next tile's global loads are in flight during the current tile).
//
// Per 32-key block and 32-query tile:
//   S^T = K Q^T        2 x v_mfma_f32_32x32x16_bf16 (sum over d = 32).  Issued "swapped" (keys
//                      as rows) so that a lane owns ONE query and 16 of the 32 keys: the
//                      softmax row max / row sum are 16 in-register ops + one lane<->lane+32
//                      exchange, no LDS.
//   P^T = exp2(S^T - m)  in registers; Q is pre-scaled by log2(e)/sqrt(D).
//   O^T += V^T P^T     2 x MFMA (sum over the 32 keys): the P^T accumulator tile is converted
//                      to bf16 and used directly as the B operand (its rows are the summed
//                      index); the matching k-permuted V^T fragments are transposed LDS reads
//                      (ds_read_b64_tr_b16) of the row-major V tile.
// O^T[d][q] keeps the query on the lane, so the online-softmax rescale is a per-lane scalar.
//
// With D = 32 the kernel is bound by the softmax VALU work (N^2 exponentials), not by the
// matrix cores: 4 MFMAs (128 pipe cycles) per 32x32 tile against 16 quarter-rate v_exp_f32 (256
// cycles) + ~60 plain VALU instructions (v_max3, packed f32 sub / sum, packed bf16 conversion).
#include "tdx_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define FA_D 32
#define FA_KT 64          // keys per staged tile
#define FA_QW 64          // queries per wave
#define FA_QB (4 * FA_QW) // queries per workgroup

// K tile rows of 64 B: chunk c of row r at c ^ ((r >> 2) & 3)  (conflict-free ds_read_b128, see conv1)
__device__ __forceinline__ int fa_sw64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }


// operand element of the matrix-core attention kernels: bf16 (the model's bf16 mode) or fp16 (BASELINE configs[4]:
// "fp16 MFMA QK^T / AV" -- 11 significand bits in P, K and V instead of 8)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
struct AttnBf16 {
    typedef bf16 T;
    typedef bf16x8 V8;
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ unsigned packp(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
struct AttnF16 {
    typedef f16 T;
    typedef f16x8 V8;
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pack_f16x2(a, b); }
    // probabilities (in [0, 1], consumed at once by the next MFMA): one v_cvt_pkrtz_f16_f32 -- the kernel is bound by
    // its vector-ALU work, round-to-nearest costs three instructions per pair
    static __device__ __forceinline__ unsigned packp(float a, float b) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
        return *reinterpret_cast<const unsigned*>(&h);
    }
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename V8>
__device__ __forceinline__ V8 fa_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(V8, r);
}

// (256, 2): two workgroups per CU = two waves per SIMD, i.e. up to 256 architectural VGPRs per lane --
// without the bound the allocator parks the accumulators in AGPRs and pays ~60 v_accvgpr moves per tile
template <typename E>
__global__ void __launch_bounds__(256, 2)
attn_fwd_mfma_kernel(const typename E::T* __restrict__ qkv, typename E::T* __restrict__ out, float* __restrict__ lse, int N, int H) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FA_KT * 64];
    unsigned char* sK = smem;                // [64 keys][32 d] bf16, swizzled 16-B chunks
    unsigned char* sV = smem + FA_KT * 64;   // [64 keys][32 d] bf16, plain rows (transposed reads)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int ld = 3 * H * FA_D;
    const typename E::T* base = qkv + (int64_t)b * N * ld;
    const int q0 = blockIdx.x * FA_QB + wave * FA_QW;

    // ---- Q fragments (B operand: col = query r, k = d), pre-scaled by log2(e)/sqrt(D)
    const float qscale = 1.4426950408889634f * rsqrtf((float)FA_D);
    typename E::V8 qf[2][2];  // [q tile][k step]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = min(q0 + qt * 32 + r, N - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Vec8<typename E::T> v;
            v.load(base + (int64_t)q * ld + h * FA_D + ks * 16 + hh * 8);
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = E::pack2(v.v[2 * e] * qscale, v.v[2 * e + 1] * qscale);
            qf[qt][ks] = __builtin_bit_cast(typename E::V8, make_uint4(w[0], w[1], w[2], w[3]));
        }
    }

    f32x16 o[2];      // O^T[d][q] per q tile: lane = query, registers = 16 of the 32 d
    float m[2], l[2]; // running max (base-2 units) and sum, per lane = per query (both halves agree)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[qt][i] = 0.f;
    }

    // staging role: 64 keys x 4 chunks of 16 B for K and for V = 512 pieces -> 2 per thread
    const int st_key = tid >> 2, st_c = tid & 3;
    uint4 kreg, vreg;
    auto load_tile = [&](int k0) {
        const int key = min(k0 + st_key, N - 1);
        const typename E::T* kp = base + (int64_t)key * ld + H * FA_D + h * FA_D + st_c * 8;
        kreg = *reinterpret_cast<const uint4*>(kp);
        vreg = *reinterpret_cast<const uint4*>(kp + H * FA_D);
    };
    // transposed-read lane geometry for V^T fragments
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int v_col = (16 * (g & 1) + 4 * tp) * 2;  // byte offset of this lane's 4 d columns

    load_tile(0);
    for (int k0 = 0; k0 < N; k0 += FA_KT) {
        __syncthreads();
        *reinterpret_cast<uint4*>(sK + fa_sw64(st_key, st_c)) = kreg;
        *reinterpret_cast<uint4*>(sV + st_key * 64 + st_c * 16) = vreg;
        __syncthreads();
        if (k0 + FA_KT < N) load_tile(k0 + FA_KT);

        auto key_block = [&](int kb, auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            // K fragments (A operand: row = key r, k = d)
            typename E::V8 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[ks] = *reinterpret_cast<const typename E::V8*>(sK + fa_sw64(kb * 32 + r, 2 * ks + hh));
            // V^T fragments for the two key sub-steps s: element j <-> key 16 s + 8 (j >> 2) + 4 hh + (j & 3)
            typename E::V8 vf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char* vp = sV + (kb * 32 + 16 * s + 4 * (g >> 1) + tq) * 64 + v_col;
                vf[s] = fa_tr_frag<typename E::V8>(vp, vp + 8 * 64);
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x16 st;
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i] = 0.f;
                st = E::mfma(kf[0], qf[qt][0], st);
                st = E::mfma(kf[1], qf[qt][1], st);
                if (TAIL) {  // keys beyond N: register i <-> key (i & 3) + 8 (i >> 2) + 4 hh
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) st[i] = -INFINITY;
                }
                // Row statistics on the vector ALU are the bottleneck of this kernel (d = 32): use the
                // 3-input max, two-wide packed f32 ops, and skip the rescale of O while the running
                // maximum of every query of the wave stays put (the common case after a few tiles).
                float mx = fmaxf(st[0], st[1]);
#pragma unroll
                for (int i = 2; i < 16; i += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[i]), st[i + 1]);  // v_max3_f32
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m[qt], mx);
                if (__builtin_amdgcn_ballot_w64(mn > m[qt]) != 0) {  // wave-uniform: some query's maximum grew
                    const float alpha = __builtin_amdgcn_exp2f(m[qt] - mn);
                    const f32x2 a2 = {alpha, alpha};
                    l[qt] *= alpha;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f32x2 t = {o[qt][2 * i], o[qt][2 * i + 1]};
                        t *= a2;  // v_pk_mul_f32
                        o[qt][2 * i] = t.x; o[qt][2 * i + 1] = t.y;
                    }
                    m[qt] = mn;
                }
                const f32x2 mn2 = {mn, mn};
                f32x2 rs2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x2 t = {st[2 * i], st[2 * i + 1]};
                    t -= mn2;  // v_pk_add_f32
                    t.x = __builtin_amdgcn_exp2f(t.x);
                    t.y = __builtin_amdgcn_exp2f(t.y);
                    rs2 += t;  // v_pk_add_f32
                    st[2 * i] = t.x; st[2 * i + 1] = t.y;
                }
                float rs = rs2.x + rs2.y;
                rs += __shfl_xor(rs, 32, 64);
                l[qt] += rs;
                // P^T as the B operand of O^T += V^T P^T: registers 8 s .. 8 s + 7 -> k step s
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const typename E::V8 pf = __builtin_bit_cast(
                        typename E::V8, make_uint4(E::packp(st[8 * s], st[8 * s + 1]), E::packp(st[8 * s + 2], st[8 * s + 3]),
                                           E::packp(st[8 * s + 4], st[8 * s + 5]), E::packp(st[8 * s + 6], st[8 * s + 7])));
                    o[qt] = E::mfma(vf[s], pf, o[qt]);
                }
            }
        };
        // the masking of keys beyond N is compiled only into the (wave-uniform) last-tile path
        if (k0 + FA_KT <= N) {
#pragma unroll
            for (int kb = 0; kb < FA_KT / 32; ++kb) key_block(kb, std::false_type{});
        } else {
#pragma unroll
            for (int kb = 0; kb < FA_KT / 32; ++kb) key_block(kb, std::true_type{});
        }
    }

    // ---- epilogue: O[q][h*D + d] = O^T[d][q] / l;  lane holds d = (i & 3) + 8 (i >> 2) + 4 hh
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 32 + r;
        if (q < N) {
            const float inv = 1.0f / l[qt];
            typename E::T* op = out + ((int64_t)b * N + q) * (H * FA_D) + h * FA_D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 v = make_uint2(E::pack2(o[qt][4 * j] * inv, o[qt][4 * j + 1] * inv),
                                           E::pack2(o[qt][4 * j + 2] * inv, o[qt][4 * j + 3] * inv));
                *reinterpret_cast<uint2*>(op + 8 * j + 4 * hh) = v;
            }
            if (hh == 0) lse[((int64_t)b * H + h) * N + q] = (m[qt] + log2f(l[qt])) * 0.6931471805599453f;
        }
    }
}

bool attn_mfma_supported(int N, int D) { return D == FA_D && N >= 128; }  // incl. the U-Net bottleneck (N = 144 at 192x64x48)

int attn_fwd_mfma_launch(const void* qkv, void* out, float* lse, int B, int N, int H, int dtype, hipStream_t st) {
    dim3 grid(ceil_div(N, FA_QB), B * H);
    if (dtype == TDX_F16)
        hipLaunchKernelGGL(attn_fwd_mfma_kernel<AttnF16>, grid, dim3(256), 0, st, (const f16*)qkv, (f16*)out, lse, N, H);
    else
        hipLaunchKernelGGL(attn_fwd_mfma_kernel<AttnBf16>, grid, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, lse, N, H);
    return tdx_launch_status();
}
