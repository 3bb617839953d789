// Matrix-core kernels of the general 3-D convolution family (the vector-ALU kernels and the layer list
// are in tdx_convg.hip), gfx950, bf16 tensors with fp32 accumulation:
//   convg_mfma_kernel<TRANSPOSED, NT>    gather / scatter^T   = forward and data gradient of the dilated (dilresnet.py:22-38),
//                                        strided (tfnet.py:185-199) and transposed (tfnet.py:201-208) convolutions
//   convg_wgrad_mfma_kernel              their weight gradients
// Same entry points, same operands as the vector-ALU kernels (weights [taps][Cin][Cout] fp32): tdx_convg_apply and
// tdx_convg_bwd_weight route bf16 tensors here.
//
// These layers have no brick structure to exploit (dilation 8 spreads the 27 taps over a 17^3 neighbourhood; stride 2 reads
// every 8th voxel), so the input operand is NOT staged: an MFMA B fragment of v_mfma_f32_32x32x16_bf16 is 8 consecutive
// channels of one voxel = one 16-B global load per lane straight from the NDHWC row of the tap's source voxel (rows of
// neighbouring taps are served by L1 / L2).  Only the weights go through LDS: per (tap, <= 128-channel slice) the fp32
// [ci][co] block is converted to bf16 and laid out as fragments, double-buffered, one barrier per stage.  The accumulator
// rows are channels in a permuted order so that a lane ends up with 16 consecutive channels of its voxel (two 16-B stores).
//
// scatter^T with stride s (transposed conv forward, strided conv data gradient): an output voxel o only receives the taps
// with (o + pad - t dil) divisible by s along every axis.  Workgroups therefore own voxels of ONE residue class o mod s, the
// divisibility test is uniform per workgroup and the dead taps (7 of 8 at stride 2) are skipped, not masked.
#include "tdx_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct ConvGM {
    int B;
    int Ei[3], Eo[3];  // grid of `in` and of `out`
    int k, stride, dil, pad;
    int clamp;         // gather only: replicate padding
    int Cin, Cout;     // channels of `in` and of `out`
    int cs;            // residue classes per axis (= stride for scatter^T, else 1)
    int cls_blocks;    // workgroups per residue class
};

#define CGM_KCH 128   // channels per weight stage
#define CGM_ROWS 256  // output voxels per workgroup

// the accumulator row rho = 8 j + 4 hh + i (register 4 j + i of lane half hh) holds channel 16 hh + 4 j + i
__device__ __forceinline__ int cgm_row_channel(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

template <bool TRANSPOSED, int NT>
__global__ void __launch_bounds__(256, 2)
convg_mfma_kernel(const bf16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, bf16* __restrict__ out,
                  ConvGM g) {
    constexpr int BN = 32 * NT;
    constexpr int WBUF = (CGM_KCH / 8) * BN * 16;  // one weight stage: [k group][channel][8] bf16
    __shared__ __attribute__((aligned(16))) unsigned char sW[2 * WBUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int cls = blockIdx.x / g.cls_blocks, blk = blockIdx.x - cls * g.cls_blocks;
    const int n0 = blockIdx.y * BN;
    const int cs = g.cs;
    const int c[3] = {cls / (cs * cs), (cls / cs) % cs, cls % cs};
    int Ec[3];  // voxels of this class per axis
#pragma unroll
    for (int a = 0; a < 3; ++a) Ec[a] = g.Eo[a] > c[a] ? (g.Eo[a] - c[a] + cs - 1) / cs : 0;
    const int64_t Mc = (int64_t)g.B * Ec[0] * Ec[1] * Ec[2];
    if ((int64_t)blk * CGM_ROWS >= Mc) return;  // uniform

    // this lane's two output voxels (M tiles mt = 0, 1 of the wave's 64 rows)
    int o[2][3], ob[2];
    bool live[2];
    int64_t orow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int64_t m = (int64_t)blk * CGM_ROWS + wave * 64 + mt * 32 + r;
        live[mt] = m < Mc;
        if (!live[mt]) m = 0;
        const int z = (int)(m % Ec[2]); m /= Ec[2];
        const int y = (int)(m % Ec[1]); m /= Ec[1];
        const int x = (int)(m % Ec[0]);
        ob[mt] = (int)(m / Ec[0]);
        o[mt][0] = c[0] + cs * x; o[mt][1] = c[1] + cs * y; o[mt][2] = c[2] + cs * z;
        orow[mt] = (((int64_t)ob[mt] * g.Eo[0] + o[mt][0]) * g.Eo[1] + o[mt][1]) * g.Eo[2] + o[mt][2];
    }

    f32x16 acc[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    const int k = g.k, ntaps = k * k * k;
    const int nkc = (g.Cin + CGM_KCH - 1) / CGM_KCH;
    // scatter^T: a tap is live for this residue class iff (c + pad - t dil) is divisible by the stride along every axis
    auto tap_live = [&](int t) -> bool {
        if (!TRANSPOSED || cs == 1) return true;
        const int tt[3] = {t / (k * k), (t / k) % k, t % k};
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int num = c[a] + g.pad - tt[a] * g.dil;
            ok = ok && (((num % cs) + cs) % cs) == 0;
        }
        return ok;
    };
    // stage = (tap, channel slice); stages are walked in order, dead taps skipped
    auto next_stage = [&](int& t, int& kc) {
        if (++kc < nkc) return;
        kc = 0;
        do { ++t; } while (t < ntaps && !tap_live(t));
    };

    // weight staging: thread (kg, c4) converts the 8 x 4 block  ci = c0 + 8 kg .. + 7,  co = n0 + 4 c4 .. + 3
    constexpr int C4 = BN / 4;
    const int s_kg = tid / C4, s_c4 = tid - s_kg * C4;
    const bool s_on = s_kg < CGM_KCH / 8;
    float4 wreg[8];
    auto stage_load = [&](int t, int kc) {
        const int c0 = kc * CGM_KCH;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = c0 + 8 * s_kg + e, co = n0 + 4 * s_c4;
            wreg[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s_on && ci < g.Cin && co < g.Cout) wreg[e] = *reinterpret_cast<const float4*>(w + ((int64_t)t * g.Cin + ci) * g.Cout + co);
        }
    };
    auto stage_store = [&](int buf) {
        if (!s_on) return;
        unsigned char* dst = sW + buf * WBUF + (s_kg * BN + 4 * s_c4) * 16;
        const float* f = reinterpret_cast<const float*>(wreg);  // [e][j]
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint4*>(dst + j * 16) = make_uint4(pack_bf16x2(f[0 + j], f[4 + j]), pack_bf16x2(f[8 + j], f[12 + j]),
                                                                 pack_bf16x2(f[16 + j], f[20 + j]), pack_bf16x2(f[24 + j], f[28 + j]));
    };

    int t = 0, kc = 0;
    while (t < ntaps && !tap_live(t)) ++t;
    if (t < ntaps) stage_load(t, kc);
    const int wslot = cgm_row_channel(r) * 16;  // the weight row this lane feeds into the MFMA
    int it = 0;
    while (t < ntaps) {
        const int buf = it & 1;
        stage_store(buf);
        __syncthreads();
        int tn = t, kn = kc;
        next_stage(tn, kn);
        if (tn < ntaps) stage_load(tn, kn);
        // source voxel of this tap for the lane's two rows
        const int tt[3] = {t / (k * k), (t / k) % k, t % k};
        const bf16* src[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            bool ok = live[mt];
            int q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (TRANSPOSED) {
                    const int num = o[mt][a] + g.pad - tt[a] * g.dil;  // divisible by the stride (tap_live), or cs == 1
                    q[a] = cs == 1 ? num : (num >= 0 ? num / cs : -1);
                } else {
                    q[a] = o[mt][a] * g.stride - g.pad + tt[a] * g.dil;
                    if (g.clamp) q[a] = min(max(q[a], 0), g.Ei[a] - 1);
                }
                ok = ok && q[a] >= 0 && q[a] < g.Ei[a];
            }
            src[mt] = ok ? in + ((((int64_t)ob[mt] * g.Ei[0] + q[0]) * g.Ei[1] + q[1]) * g.Ei[2] + q[2]) * g.Cin + kc * CGM_KCH + 8 * hh
                         : nullptr;
        }
        const int kcs = min(CGM_KCH, g.Cin - kc * CGM_KCH);  // channels of this slice
        const unsigned char* wb = sW + buf * WBUF + hh * (BN * 16) + wslot;

        for (int ks = 0; ks * 16 < kcs; ++ks) {
            const bool kin = ks * 16 + 8 * hh < kcs;  // (a slice of 8 (mod 16) channels ends in a half K step)
            bf16x8 xf[2], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                uint4 u = make_uint4(0, 0, 0, 0);
                if (src[mt] != nullptr && kin) u = *reinterpret_cast<const uint4*>(src[mt] + ks * 16);
                xf[mt] = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wb + (2 * ks * BN + nt * 32) * 16));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        t = tn; kc = kn;
        ++it;
    }

    // ---- epilogue: lane (r, hh) holds channels n0 + 32 nt + 16 hh + (0..15) of its voxel in registers 0..15
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = n0 + 32 * nt + 16 * hh;
        float bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = (bias != nullptr && ch + i < g.Cout) ? bias[ch + i] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (!live[mt]) continue;
            bf16* dst = out + orow[mt] * g.Cout + ch;
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                if (ch + 8 * h8 >= g.Cout) continue;
                Vec8<bf16> v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v.v[i] = acc[nt][mt][8 * h8 + i] + bv[8 * h8 + i];
                v.store(dst + 8 * h8);
            }
        }
    }
}

int convg_mfma_apply(const void* in, const float* w, const float* bias, void* out, int B, const int* Ei, const int* Eo, int Cin,
                     int Cout, int k, int stride, int dil, int pad, int replicate, int transposed, hipStream_t st) {
    ConvGM g;
    g.B = B;
    for (int a = 0; a < 3; ++a) { g.Ei[a] = Ei[a]; g.Eo[a] = Eo[a]; }
    g.k = k; g.stride = stride; g.dil = dil; g.pad = pad; g.clamp = replicate; g.Cin = Cin; g.Cout = Cout;
    g.cs = transposed ? stride : 1;
    int64_t mc = B;  // class 0 has the most voxels
    for (int a = 0; a < 3; ++a) mc *= (Eo[a] + g.cs - 1) / g.cs;
    g.cls_blocks = ceil_div(mc, CGM_ROWS);
    const int nt = Cout > 32 ? 2 : 1;
    dim3 grid((unsigned)(g.cls_blocks * g.cs * g.cs * g.cs), (unsigned)ceil_div(Cout, 32 * nt));
#define CGM_GO(TR, NTV) \
    hipLaunchKernelGGL((convg_mfma_kernel<TR, NTV>), grid, dim3(256), 0, st, (const bf16*)in, w, bias, (bf16*)out, g)
    if (transposed) { if (nt == 2) CGM_GO(true, 2); else CGM_GO(true, 1); }
    else            { if (nt == 2) CGM_GO(false, 2); else CGM_GO(false, 1); }
#undef CGM_GO
    return tdx_launch_status();
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW[t][ci][co] = sum over output voxels o of in[src(o, t)][ci] dy[o][co]: per tap a GEMM with K = voxels, both operands
// voxel-major, so every fragment is a transposed LDS read (ds_read_b64_tr_b16, as tdx_conv1_mfma.hip's weight gradient).
// One workgroup = (tap, 64 ci x 64 co tile, every nsplit-th chunk of 256 output voxels); thread i stages row i of a chunk
// (its source voxel for the tap is computed once per chunk), the next chunk's rows are in flight during the MFMAs.
// Partial tiles are merged with fp32 atomics; dW / dbias must be zero on entry.
#define CGW_PLANE (CGM_ROWS * 64)  // one [256 voxels][32 channels] bf16 plane

__device__ __forceinline__ bf16x8 cgm_tr_frag(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 rr = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, rr);
}

__global__ void __launch_bounds__(256, 2)
convg_wgrad_mfma_kernel(const bf16* __restrict__ in, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                        ConvGM g, int nsplit, int n_ci_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * CGW_PLANE];
    unsigned char* sX = smem;
    unsigned char* sG = smem + 2 * CGW_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 64, co0 = (tile / n_ci_tiles) * 64;
    const int tap = blockIdx.y, k = g.k;
    const int tt[3] = {tap / (k * k), (tap / k) % k, tap % k};
    const int grp = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int col_off = (16 * (grp & 1) + 4 * p4) * 2;
    const int kh = grp >> 1;

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[0][0][i] = acc[0][1][i] = acc[1][0][i] = acc[1][1][i] = 0.f;
        accb[0][i] = accb[1][i] = 0.f;
    }
    const bool do_bias = dbias != nullptr && tap == 0 && ci0 == 0;
    const int64_t rows = (int64_t)g.B * g.Eo[0] * g.Eo[1] * g.Eo[2];
    const int64_t nchunks = (rows + CGM_ROWS - 1) / CGM_ROWS;

    uint4 xr[8], gr[8];
    auto load_chunk = [&](int64_t ch) {
        int64_t m = ch * CGM_ROWS + tid;
        const bool live = m < rows;
        const bf16* gp = dy + m * g.Cout + co0;
        if (!live) m = 0;
        const int z = (int)(m % g.Eo[2]); m /= g.Eo[2];
        const int y = (int)(m % g.Eo[1]); m /= g.Eo[1];
        const int x = (int)(m % g.Eo[0]);
        const int b = (int)(m / g.Eo[0]);
        const int oo[3] = {x, y, z};
        int qq[3];
        bool ok = live;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            qq[a] = oo[a] * g.stride - g.pad + tt[a] * g.dil;
            if (g.clamp) qq[a] = min(max(qq[a], 0), g.Ei[a] - 1);
            ok = ok && qq[a] >= 0 && qq[a] < g.Ei[a];
        }
        const bf16* xp = in + ((((int64_t)b * g.Ei[0] + qq[0]) * g.Ei[1] + qq[1]) * g.Ei[2] + qq[2]) * g.Cin + ci0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xr[i] = gr[i] = make_uint4(0, 0, 0, 0);
            // a row whose source voxel is outside contributes nothing: its x piece is zero (dy still feeds the bias sum)
            if (ok && ci0 + 8 * i < g.Cin) xr[i] = *reinterpret_cast<const uint4*>(xp + 8 * i);
            if (live && co0 + 8 * i < g.Cout) gr[i] = *reinterpret_cast<const uint4*>(gp + 8 * i);
        }
    };
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    if (split < nchunks) load_chunk(split);
    for (int64_t ch = split; ch < nchunks; ch += nsplit) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // piece i of row tid: plane i >> 2, 16-B chunk i & 3 of the plane's 64-B row
            *reinterpret_cast<uint4*>(sX + (i >> 2) * CGW_PLANE + tid * 64 + (i & 3) * 16) = xr[i];
            *reinterpret_cast<uint4*>(sG + (i >> 2) * CGW_PLANE + tid * 64 + (i & 3) * 16) = gr[i];
        }
        __syncthreads();
        if (ch + nsplit < nchunks) load_chunk(ch + nsplit);
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // the wave reduces rows [64 wave, 64 wave + 64): 4 K steps of 16 voxels
            const int row = wave * 64 + 16 * s + 8 * kh + q4;
            bf16x8 af[2], bfv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned char* ap = sX + m * CGW_PLANE + row * 64 + col_off;
                af[m] = cgm_tr_frag(ap, ap + 4 * 64);
                const unsigned char* bp = sG + m * CGW_PLANE + row * 64 + col_off;
                bfv[m] = cgm_tr_frag(bp, bp + 4 * 64);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfv[n], acc[m][n], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int n = 0; n < 2; ++n) accb[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfv[n], accb[n], 0, 0, 0);
            }
        }
    }
    const int r = lane & 31, hh = lane >> 5;
    float* dwt = dw + (int64_t)tap * g.Cin * g.Cout;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = ci0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh, co = co0 + n * 32 + r;
                if (ci < g.Cin && co < g.Cout) atomicAdd(&dwt[(int64_t)ci * g.Cout + co], acc[m][n][i]);
            }
    if (do_bias && hh == 0) {  // every row of accb is the column sum; row 0 sits in register 0 of lane half 0
#pragma unroll
        for (int n = 0; n < 2; ++n)
            if (co0 + n * 32 + r < g.Cout) atomicAdd(&dbias[co0 + n * 32 + r], accb[n][0]);
    }
}

int convg_mfma_bwd_weight(const void* in, const void* dy, float* dw, float* dbias, int B, const int* Ei, const int* Eo, int Cin,
                          int Cout, int k, int stride, int dil, int pad, int replicate, hipStream_t st) {
    ConvGM g;
    g.B = B;
    for (int a = 0; a < 3; ++a) { g.Ei[a] = Ei[a]; g.Eo[a] = Eo[a]; }
    g.k = k; g.stride = stride; g.dil = dil; g.pad = pad; g.clamp = replicate; g.Cin = Cin; g.Cout = Cout;
    g.cs = 1; g.cls_blocks = 0;
    const int n_ci = ceil_div(Cin, 64), n_co = ceil_div(Cout, 64);
    const int64_t rows = (int64_t)B * Eo[0] * Eo[1] * Eo[2];
    const int64_t nchunks = (rows + CGM_ROWS - 1) / CGM_ROWS;
    // enough workgroups for four rounds of the chip, at least four chunks per workgroup
    const int64_t base = (int64_t)k * k * k * n_ci * n_co;
    int64_t nsplit = (2048 + base - 1) / base;
    if (nsplit > nchunks / 4) nsplit = nchunks / 4;
    if (nsplit < 1) nsplit = 1;
    dim3 grid((unsigned)(n_ci * n_co * nsplit), (unsigned)(k * k * k));
    hipLaunchKernelGGL(convg_wgrad_mfma_kernel, grid, dim3(256), 0, st, (const bf16*)in, (const bf16*)dy, dw, dbias, g, (int)nsplit, n_ci);
    return tdx_launch_status();
}
