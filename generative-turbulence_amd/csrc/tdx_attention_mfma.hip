// bf16 / fp16 MFMA flash attention (forward) for long token sequences (BASELINE config 5:
// N = 96*32*24 = 73 728 voxels, 4 heads x 32).  gfx950.
//
// qkv is the NDHWC to_qkv output [B][N][3*H*D] (q | k | v thirds, head-major), D = 32.
// One wave owns 32 queries of one (b, h); a workgroup = 8 waves = 256 queries, and walks the
// keys in tiles of 64 staged through LDS (K and V, 4 KB each, shared by the 8 waves; the next
// tile's global loads are in flight during the current tile).
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
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define FA_D 32
#define FA_KT 64          // keys per staged tile
#define FA_QW 32          // queries per wave (one 32-query tile: <= 128 registers, four waves per SIMD)
#define FA_THREADS 512    // 8 waves
#define FA_QB (8 * FA_QW) // queries per workgroup

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
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float unpack_lo(unsigned w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float unpack_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
    static constexpr float lazy = 32.0f;
    static constexpr float headroom = 0.0f;
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
    static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float unpack_lo(unsigned w) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, w)[0];
    }
    static __device__ __forceinline__ float unpack_hi(unsigned w) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return (float)__builtin_bit_cast(h2, w)[1];
    }
    static constexpr float lazy = 15.0f;  // 2^15 < 65504
    // fp16's allowance is narrow: a segment may START with m up to 2^3 above its first block's maximum when that is what
    // lets the wave run without the check (probabilities then begin at 2^-3; what fp16 flushes to zero moves from 2^-24
    // to 2^-21 of the maximum)
    static constexpr float headroom = 3.0f;
};

template <typename V8>
__device__ __forceinline__ V8 fa_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(V8, r);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward kernel, round 5.  What bound the round-3 kernel (0.21 of the matrix peak at N = 73 728): per 32 x 32 score tile
// 4 MFMAs (128 matrix-pipe cycles) against ~100 vector instructions -- 16 v_exp_f32 (8 issue cycles each on gfx950), 16
// two-wide packed f32 adds (v_pk_add_f32: ~13 cycles each beside MFMAs, MI355X_MICROARCH.md per-instruction table:
// slower than two plain adds), the max chain with its lane <-> lane + 32 exchange, 16 accumulator zero moves, packed
// converts -- and a grid of 1152 equal workgroups on 512 slots (2.25 rounds run as 3).  Now:
//   * the running maximum is LAZY and enters through the MFMA's C operand: S^T - m comes out of the matrix pipe
//     (C = -m broadcast, 16 registers per query tile kept up to date), so there is no subtraction and no zeroing; m is only
//     raised when some score of the tile exceeds it by more than an allowance (wave-uniform branch, rare after the first
//     tiles): probabilities are then <= 2^allowance instead of <= 1 -- exact in floating point, fp32 accumulators;
//   * the row sums l = sum_k P come from the matrix pipe too (a 0 / 1 selector x P^T, two v_mfma_f32_16x16x32 per tile
//     and 4 accumulator registers, on a pipe that idled 60 % of the time): no vector adds, no cross-lane exchange, and l
//     is the sum of the ROUNDED probabilities the numerator uses.  The P^T fragment of the 32x32x16 product (lane = query
//     lane & 31, 8 keys of half lane >> 5) read as the B operand of a 16x16x32 product is column n = lane & 15, k group
//     lane >> 4: column n collects query n (k groups 0, 2) and query n + 16 (k groups 1, 3), so selector row 0 = ones on
//     k groups 0 and 2, row 1 = ones on k groups 1 and 3: lane n < 16 ends up with l(query n), l(query n + 16) in its
//     first two accumulator registers;
//   * whether a tile outgrew the allowance is decided by 8 v_max3_i32 + one compare on the score bit patterns -- and not
//     at all for a wave whose queries cannot get there: |s| <= |q| max_k |k| (Cauchy-Schwarz; the key norms' maximum per
//     head comes from a pre-pass over K, 10 us), so once bound - m <= allowance for every query of the wave the loop runs
//     without the check (9 of its ~60 issue slots per tile; with gaussian inputs from the first tile on);
//   * per tile the vector ALU is left with 16 v_exp_f32 and 8 packed converts (+ the check where it is needed);
//   * stream-K schedule: a persistent grid (2 workgroups per CU) splits the linear (query block, key tile) iteration
//     space evenly; a block whose key range straddles two workgroups is finished by a small merge kernel from the
//     partial (O, m, l) both wrote.  XCD x owns a contiguous eighth of the space (one head's K / V per L2).
// (allowance per operand type, E::lazy(): probabilities must stay finite in the operand format -- fp16 tops out at 2^16 --
// and their sums in fp32)


struct AttnSkArgs {
    int N, H, nqb, ntile;      // tokens, heads, query blocks per (b, h), key tiles per block
    long long total;           // iterations = blocks * ntile
    int chunk, nwg;            // iterations per workgroup, workgroups
    float* part;               // [nwg][2 slots][FA_QB][34] f32 partials (O[32], m, l), or nullptr when chunk == ntile
    const float* kmax2;        // [B * H] max_k |k|^2 of each head's keys (attn_kmax_kernel), or nullptr: always check
};
#define FA_PART_STRIDE (FA_QB * 34)

template <typename E>
__global__ void __launch_bounds__(FA_THREADS, 4)
attn_fwd_mfma_kernel(const typename E::T* __restrict__ qkv, typename E::T* __restrict__ out, float* __restrict__ lse, AttnSkArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FA_KT * 64];
    unsigned char* sK = smem;                // [64 keys][32 d], swizzled 16-B chunks
    unsigned char* sV = smem + FA_KT * 64;   // [64 keys][32 d], plain rows (transposed reads)
    using V8 = typename E::V8;
    using T = typename E::T;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int N = a.N, H = a.H;
    const int ld = 3 * H * FA_D;
    // XCD x (= hardware workgroup id mod 8) owns the x-th contiguous eighth of the iteration space
    int w = blockIdx.x;
    if ((a.nwg & 7) == 0) w = (w & 7) * (a.nwg >> 3) + (w >> 3);
    long long it = (long long)w * a.chunk;
    const long long it_end = it + a.chunk < a.total ? it + a.chunk : a.total;

    const float qscale = 1.4426950408889634f * rsqrtf((float)FA_D);
    // staging role: 64 keys x 4 chunks of 16 B, K by the first 256 threads, V by the other 256
    const int st_key = (tid & 255) >> 2, st_c = tid & 3, st_v = tid >> 8;
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int v_col = (16 * (g & 1) + 4 * tp) * 2;
    V8 sel;  // A operand of the row-sum product: A[m = lane & 15][k group = lane >> 4]
    {
        const unsigned one2 = (lane == 0 || lane == 32 || lane == 17 || lane == 49) ? E::pack2(1.0f, 1.0f) : 0u;
        sel = __builtin_bit_cast(V8, make_uint4(one2, one2, one2, one2));
    }

    while (it < it_end) {
        const int blk = (int)(it / a.ntile);
        const int t0 = (int)(it - (long long)blk * a.ntile);
        const int t1 = (int)((long long)a.ntile - t0 <= it_end - it ? a.ntile : t0 + (it_end - it));
        const int bh = blk / a.nqb, qb = blk - bh * a.nqb;
        const int b = bh / H, h = bh - b * H;
        const T* base = qkv + (int64_t)b * N * ld;
        const int q0 = qb * FA_QB + wave * FA_QW;

        // ---- Q fragments (B operand: col = query r, k = d), pre-scaled by log2(e)/sqrt(D)
        V8 qf[2];
        {
            const int q = min(q0 + r, N - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Vec8<T> v;
                v.load(base + (int64_t)q * ld + h * FA_D + ks * 16 + hh * 8);
                unsigned wv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[e] = E::pack2(v.v[2 * e] * qscale, v.v[2 * e + 1] * qscale);
                qf[ks] = __builtin_bit_cast(V8, make_uint4(wv[0], wv[1], wv[2], wv[3]));
            }
        }
        // |q|^2 of the lane's (rounded, pre-scaled) query: 16 of its 32 components here, the rest in lane ^ 32
        float bound = INFINITY;  // upper bound of every score of this query against this head's keys (log2 units)
        if (a.kmax2 != nullptr) {
            float q2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 u = __builtin_bit_cast(uint4, qf[ks]);
                const unsigned wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = E::unpack_lo(wv[e]), hi = E::unpack_hi(wv[e]);
                    q2 = __builtin_fmaf(lo, lo, __builtin_fmaf(hi, hi, q2));
                }
            }
            q2 += __shfl_xor(q2, 32, 64);
            bound = 1.001f * sqrtf(q2 * a.kmax2[bh]) + 1e-3f;
        }
        f32x16 o, negm;  // O^T[d][q]; -m broadcast over the 16 score registers
        f32x4 ls;        // row sums: lanes 0..15, [0] = query lane, [1] = query lane + 16
        float m;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ls[i] = 0.f;

        uint4 sreg;
        auto load_tile = [&](int k0) {
            const int key = min(k0 + st_key, N - 1);
            sreg = *reinterpret_cast<const uint4*>(base + (int64_t)key * ld + (1 + st_v) * H * FA_D + h * FA_D + st_c * 8);
        };
        auto stage = [&](int t) {  // tile t: registers -> LDS, next tile's loads in flight
            __syncthreads();
            if (st_v == 0) *reinterpret_cast<uint4*>(sK + fa_sw64(st_key, st_c)) = sreg;
            else *reinterpret_cast<uint4*>(sV + st_key * 64 + st_c * 16) = sreg;
            __syncthreads();
            if (t + 1 < t1) load_tile((t + 1) * FA_KT);
        };
        // raise the running maximum by what the scores `mx` (relative to the stale one) ask for; everything accumulated
        // under the old maximum shrinks by alpha.  Rare; it sits on the loop's retry edge, not in the
        // straight-line tile code: an in-place update of the 16-register MFMA tuples inside a branch there made the
        // compiler copy o / negm (18 v_mov_b64) on the common path of every tile.
        auto raise = [&](float mx) {
            const float d = fmaxf(fmaxf(mx, __shfl_xor(mx, 32, 64)), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            m += d;
            ls[0] *= alpha;                                // lane n < 16: query n ...
            ls[1] *= __shfl(alpha, (lane + 16) & 63, 64);  // ... and query n + 16
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o[i] *= alpha;
                negm[i] -= d;
            }
            return d;
        };
        auto row_max = [&](const f32x16& st) {  // exact (the rare paths and the segment's first block)
            float mx = st[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[i]);
            return mx;
        };
        // "does any score exceed the allowance?" needs no float maximum: for positive floats the order of the bit patterns
        // as signed integers is the order of the values, and any negative float is a negative integer -- so the integer
        // maximum of the patterns exceeds bits(allowance) exactly when some score does.  v_max3_i32 straight on the MFMA
        // result: no canonicalising v_max in front (fmaxf on matrix-pipe outputs gets one per operand), no asm hazard nops.
        auto row_max_bits = [&](const f32x16& st) {
            int mx = max(max(__float_as_int(st[0]), __float_as_int(st[1])), __float_as_int(st[2]));
#pragma unroll
            for (int i = 3; i < 15; i += 2) mx = max(max(mx, __float_as_int(st[i])), __float_as_int(st[i + 1]));
            return max(mx, __float_as_int(st[15]));
        };
        auto frags = [&](int kb, V8 (&kf)[2], V8 (&vf)[2]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                kf[ks] = *reinterpret_cast<const V8*>(sK + fa_sw64(kb * 32 + r, 2 * ks + hh));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const unsigned char* vp = sV + (kb * 32 + 16 * s2 + 4 * (g >> 1) + tq) * 64 + v_col;
                vf[s2] = fa_tr_frag<V8>(vp, vp + 8 * 64);
            }
        };
        auto accumulate = [&](f32x16& st, const V8 (&vf)[2]) {  // P = 2^st; O^T += V^T P^T; l += sum P
#pragma unroll
            for (int i = 0; i < 16; ++i) st[i] = __builtin_amdgcn_exp2f(st[i]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const V8 pf = __builtin_bit_cast(
                    V8, make_uint4(E::packp(st[8 * s2], st[8 * s2 + 1]), E::packp(st[8 * s2 + 2], st[8 * s2 + 3]),
                                   E::packp(st[8 * s2 + 4], st[8 * s2 + 5]), E::packp(st[8 * s2 + 6], st[8 * s2 + 7])));
                o = E::mfma(vf[s2], pf, o);
                ls = E::mfma16(sel, pf, ls);  // row sums of the rounded probabilities
            }
        };

        load_tile(t0 * FA_KT);
        stage(t0);
        {   // the maximum over the segment's first 32 keys starts m (key t0 * 64 < N: finite)
            f32x16 s0;
#pragma unroll
            for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                s0 = E::mfma(*reinterpret_cast<const V8*>(sK + fa_sw64(r, 2 * ks + hh)), qf[ks], s0);
            if (t0 * FA_KT + 32 > N) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t0 * FA_KT + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) s0[i] = -INFINITY;
            }
            float mx = row_max(s0);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (E::headroom > 0.f) {  // (bound = inf without the key norms: no lift)
                const float lift = bound - E::lazy + 0.1f - mx;  // what m lacks for the check-free loop
                if (__builtin_amdgcn_ballot_w64(lift > E::headroom) == 0) mx += fmaxf(lift, 0.f);
            }
            m = mx;
#pragma unroll
            for (int i = 0; i < 16; ++i) negm[i] = -mx;
        }

        // ---- main loop over the 32-key blocks j of the tiles that lie wholly below N: no masking, and no in-place
        // update on the straight path (four waves per SIMD hide each other's matrix-pipe and LDS latencies)
        const int t_full = min(t1, N / FA_KT);  // tiles [t0, t_full) have all 64 keys
        int j = 2 * t0;
        bool staged = true;  // tile j >> 1 is in LDS
        // can some query of this wave still meet a score above its stale maximum + allowance?
        bool check = __builtin_amdgcn_ballot_w64(bound - m > E::lazy - 0.05f) != 0;
        while (check && j < 2 * t_full) {  // loop 1: looks at every block's scores
            if (!staged) { stage(j >> 1); staged = true; }
            V8 kf[2], vf[2];
            frags(j & 1, kf, vf);
            f32x16 st = E::mfma(kf[0], qf[0], negm);  // S^T - m straight from the matrix pipe
            st = E::mfma(kf[1], qf[1], st);
            if (__builtin_amdgcn_ballot_w64(row_max_bits(st) > __float_as_int(E::lazy)) != 0) {
                raise(row_max(st));
                check = __builtin_amdgcn_ballot_w64(bound - m > E::lazy - 0.05f) != 0;
                continue;  // the same block again, under the new maximum (or in loop 2)
            }
            accumulate(st, vf);
            ++j;
            staged = (j & 1) != 0;
        }
        for (; j < 2 * t_full; ++j) {  // loop 2: no score of the rest can outgrow the allowance
            if (!staged) stage(j >> 1);
            V8 kf[2], vf[2];
            frags(j & 1, kf, vf);
            f32x16 st = E::mfma(kf[0], qf[0], negm);
            st = E::mfma(kf[1], qf[1], st);
            accumulate(st, vf);
            staged = (j & 1) == 0;
        }
        // ---- the block's last tile when N is not a multiple of 64: keys beyond N masked, maxima exact
        if (t_full < t1) {
            if (t_full != t0) stage(t_full);  // (the segment's first tile is in LDS already)
            const int k0 = t_full * FA_KT;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                V8 kf[2], vf[2];
                frags(kb, kf, vf);
                f32x16 st = E::mfma(kf[0], qf[0], negm);
                st = E::mfma(kf[1], qf[1], st);
#pragma unroll
                for (int i = 0; i < 16; ++i)  // register i <-> key (i & 3) + 8 (i >> 2) + 4 hh
                    if (k0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= N) st[i] = -INFINITY;
                const float mx = row_max(st);
                if (__builtin_amdgcn_ballot_w64(mx > 0.f) != 0) {
                    const float d = raise(mx);
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[i] -= d;
                }
                accumulate(st, vf);
            }
        }

        // ---- segment epilogue; lane holds d = (i & 3) + 8 (i >> 2) + 4 hh of query r
        const float l0 = __shfl(ls[0], r & 15, 64), l1 = __shfl(ls[1], r & 15, 64);
        const float lq = r < 16 ? l0 : l1;  // l of the lane's query: from lane r & 15, register r >> 4
        const int q = q0 + r;
        if (t0 == 0 && t1 == a.ntile) {  // the whole key range: final result
            if (q < N) {
                const float inv = 1.0f / lq;
                T* op = out + ((int64_t)b * N + q) * (H * FA_D) + h * FA_D;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint2 v = make_uint2(E::pack2(o[4 * jj] * inv, o[4 * jj + 1] * inv),
                                               E::pack2(o[4 * jj + 2] * inv, o[4 * jj + 3] * inv));
                    *reinterpret_cast<uint2*>(op + 8 * jj + 4 * hh) = v;
                }
                if (hh == 0) lse[((int64_t)b * H + h) * N + q] = (m + log2f(lq)) * 0.6931471805599453f;
            }
        } else {
            // part of the key range: (O, m, l) relative to this segment's m; slot 0 = the block began in an earlier
            // workgroup (this workgroup's first segment), slot 1 = it continues in the next one (its last segment)
            float* pq = a.part + ((size_t)w * 2 + (t0 == 0 ? 1 : 0)) * FA_PART_STRIDE + (size_t)(wave * FA_QW + r) * 34;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float* pd = pq + 8 * jj + 4 * hh;  // 136-B rows: 8-B aligned
                *reinterpret_cast<float2*>(pd) = make_float2(o[4 * jj], o[4 * jj + 1]);
                *reinterpret_cast<float2*>(pd + 2) = make_float2(o[4 * jj + 2], o[4 * jj + 3]);
            }
            if (hh == 0) { pq[32] = m; pq[33] = lq; }
        }
        it += t1 - t0;  // (the next segment's stage() opens with a barrier: its LDS stores wait for this one's last reads)
    }
}

// max_k |k|^2 per (b, h): one pass over K, block maxima merged by integer atomics on the (non-negative) float patterns.
template <typename T>
__global__ void __launch_bounds__(256)
attn_kmax_kernel(const T* __restrict__ qkv, float* __restrict__ kmax2, int N, int H) {
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int ld = 3 * H * FA_D;
    const T* base = qkv + (int64_t)b * N * ld + H * FA_D + h * FA_D;
    float mx = 0.f;
    for (int key = blockIdx.x * 64 + (threadIdx.x >> 2); key < N; key += gridDim.x * 64) {
        Vec8<T> v;
        v.load(base + (int64_t)key * ld + (threadIdx.x & 3) * 8);
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 = __builtin_fmaf(v.v[e], v.v[e], s2);
        s2 += __shfl_xor(s2, 1, 64);
        s2 += __shfl_xor(s2, 2, 64);
        mx = fmaxf(mx, s2);
    }
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(kmax2) + bh, __float_as_int(mx));
}

// Blocks whose key range was split over workgroups: combine the partial (O, m, l) of every workgroup that touched them.
template <typename E>
__global__ void __launch_bounds__(FA_QB)
attn_fwd_merge_kernel(typename E::T* __restrict__ out, float* __restrict__ lse, AttnSkArgs a) {
    const int blk = blockIdx.x;
    const long long lo = (long long)blk * a.ntile, hi = lo + a.ntile;
    const int w_lo = (int)(lo / a.chunk), w_hi = (int)((hi - 1) / a.chunk);
    if (w_lo == w_hi) return;  // one workgroup had the whole block and wrote the result
    const int bh = blk / a.nqb, qb = blk - bh * a.nqb;
    const int b = bh / a.H, h = bh - b * a.H;
    const int ql = threadIdx.x, q = qb * FA_QB + ql;
    if (q >= a.N) return;
    float mt = -INFINITY;
    for (int w = w_lo; w <= w_hi; ++w)
        mt = fmaxf(mt, a.part[((size_t)w * 2 + (w == w_lo ? 1 : 0)) * FA_PART_STRIDE + (size_t)ql * 34 + 32]);
    float acc[FA_D], l = 0.f;
#pragma unroll
    for (int d = 0; d < FA_D; ++d) acc[d] = 0.f;
    for (int w = w_lo; w <= w_hi; ++w) {
        const float* pq = a.part + ((size_t)w * 2 + (w == w_lo ? 1 : 0)) * FA_PART_STRIDE + (size_t)ql * 34;
        const float sc = exp2f(pq[32] - mt);
        l += sc * pq[33];
#pragma unroll
        for (int d = 0; d < FA_D; d += 2) {
            const float2 v = *reinterpret_cast<const float2*>(pq + d);
            acc[d] += sc * v.x;
            acc[d + 1] += sc * v.y;
        }
    }
    const float inv = 1.0f / l;
    typename E::T* op = out + ((int64_t)b * a.N + q) * (a.H * FA_D) + h * FA_D;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint4*>(op + 8 * j) =
            make_uint4(E::pack2(acc[8 * j] * inv, acc[8 * j + 1] * inv), E::pack2(acc[8 * j + 2] * inv, acc[8 * j + 3] * inv),
                       E::pack2(acc[8 * j + 4] * inv, acc[8 * j + 5] * inv), E::pack2(acc[8 * j + 6] * inv, acc[8 * j + 7] * inv));
    lse[((int64_t)b * a.H + h) * a.N + q] = (mt + log2f(l)) * 0.6931471805599453f;
}

bool attn_mfma_supported(int N, int D) { return D == FA_D && N >= 128; }  // incl. the U-Net bottleneck (N = 144 at 192x64x48)

void* tdx_scratch_ptr();
size_t tdx_scratch_bytes();

// Persistent stream-K schedule when the launch is big enough to care about whole rounds of workgroups (and the stream's
// scratch arena can hold the partials: 2 slots x 512 workgroups x 34 816 B = 35.7 MB behind its 64-B zero block);
// otherwise one workgroup per query block.
int attn_fwd_mfma_launch(const void* qkv, void* out, float* lse, int B, int N, int H, int dtype, hipStream_t st) {
    AttnSkArgs a;
    a.N = N; a.H = H;
    a.nqb = ceil_div(N, FA_QB);
    a.ntile = ceil_div(N, FA_KT);
    const int nblk = B * H * a.nqb;
    a.total = (long long)nblk * a.ntile;
    const int slots = 512;  // 256 CUs x 2 workgroups
    const size_t head = 64 + 1024;  // zero block, then max_k |k|^2 of up to 256 (b, h) pairs
    const size_t need = head + (size_t)slots * 2 * FA_PART_STRIDE * sizeof(float);
    const char* env = getenv("TDX_ATTN_STREAMK");
    const bool want = !(env && env[0] == '0') && nblk > slots && (nblk % slots) != 0 && a.ntile >= 16;
    if (want && tdx_scratch_ptr() != nullptr && tdx_scratch_bytes() >= need) {
        a.nwg = slots;
        a.chunk = (int)((a.total + slots - 1) / slots);
        a.part = reinterpret_cast<float*>((char*)tdx_scratch_ptr() + head);
    } else {
        a.nwg = nblk;
        a.chunk = a.ntile;
        a.part = nullptr;
    }
    a.kmax2 = nullptr;
    const char* envb = getenv("TDX_ATTN_BOUND");
    if (!(envb && envb[0] == '0') && tdx_scratch_ptr() != nullptr && tdx_scratch_bytes() >= head && B * H <= 256 && a.ntile >= 8) {
        float* km = reinterpret_cast<float*>((char*)tdx_scratch_ptr() + 64);
        if (tdx_zero_async(km, (size_t)B * H * sizeof(float), st) != TDX_OK) return TDX_EINVAL;
        const dim3 kg(min(ceil_div(N, 64), 64), B * H);
        if (dtype == TDX_F16) hipLaunchKernelGGL(attn_kmax_kernel<f16>, kg, dim3(256), 0, st, (const f16*)qkv, km, N, H);
        else hipLaunchKernelGGL(attn_kmax_kernel<bf16>, kg, dim3(256), 0, st, (const bf16*)qkv, km, N, H);
        a.kmax2 = km;
    }
    if (dtype == TDX_F16)
        hipLaunchKernelGGL(attn_fwd_mfma_kernel<AttnF16>, dim3(a.nwg), dim3(FA_THREADS), 0, st, (const f16*)qkv, (f16*)out, lse, a);
    else
        hipLaunchKernelGGL(attn_fwd_mfma_kernel<AttnBf16>, dim3(a.nwg), dim3(FA_THREADS), 0, st, (const bf16*)qkv, (bf16*)out, lse, a);
    if (a.part != nullptr) {
        if (dtype == TDX_F16)
            hipLaunchKernelGGL(attn_fwd_merge_kernel<AttnF16>, dim3(nblk), dim3(FA_QB), 0, st, (f16*)out, lse, a);
        else
            hipLaunchKernelGGL(attn_fwd_merge_kernel<AttnBf16>, dim3(nblk), dim3(FA_QB), 0, st, (bf16*)out, lse, a);
    }
    return tdx_launch_status();
}
