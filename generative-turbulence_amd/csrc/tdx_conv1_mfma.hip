// bf16 MFMA 1x1x1 convolution (per-voxel channel GEMM) and its weight gradient for gfx950.
//
// These layers are HBM-bound (4-174 FLOP/B): the job of the kernels is to stream the
// activation once in full 128-B lines, keep many loads in flight (small LDS footprint ->
// 4 workgroups per CU) and never let the arithmetic show, which at 1x1-conv sizes takes the
// matrix cores (the vector-ALU version ran at ~1/10 of the HBM rate).
//
// forward:  y[r, n] = bias[n] + sum_k [x1|x2][r, k] w[k, n] (+ add[r, n])
//   workgroup = 256 rows x BN = 32*NT columns; K walked in 32-channel slices.  The x slice
//   ([256][32] bf16, 64-B rows) and the weight slice (f32 [k][n] in memory, converted and
//   transposed to [n][32] bf16 while staging) go through LDS; 16-B chunks are XOR-swizzled
//   with (row >> 2) & 3 so that ds_read_b128 fragment reads are conflict-free.  The MFMA is
//   issued as D^T = W^T X^T (weights as the A operand): each lane then owns one voxel row
//   and 4 consecutive channels per accumulator quad, which packs to 8-B LDS writes of an
//   output tile that is finally stored with 16-B coalesced rows (+ bias, + residual).
//
// weight gradient:  dw[ci, co] = sum_r x[r, ci] dy[r, co]   (TN GEMM over rows)
//   both operands are row(K)-major, so fragments are transposed LDS reads
//   (ds_read_b64_tr_b16) from [256][32]-channel planes (64-B rows: conflict-free).  Each wave
//   reduces its own 64 rows of every 256-row chunk into a private (32*MT) x (32*NT) tile;
//   tiles are merged with f32 atomics.
#include "tdx_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define C1M_ROWS 256
#define C1M_KC 32

bool conv1_mfma_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % C1M_KC) == 0 && (C2 % C1M_KC) == 0 && (Cout % 32) == 0;
}

// 64-B rows, 4 chunks of 16 B; chunk c of row r lives at chunk position c ^ ((r >> 2) & 3)
__device__ __forceinline__ int sw64(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }
// 128-B rows (64 bf16), 8 chunks; chunk c of row r at c ^ (r & 7)
__device__ __forceinline__ int sw128(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// `add` through a normalisation: with gn.stats != nullptr the addend is silu(GroupNorm(add)) -- the tail of a ResnetBlock
// with a projected skip, y = silu(GN(h2)) + conv1x1(x) (reference ddpm.py:176,197), in ONE pass: the skip tensor is never
// written or re-read.  Same arithmetic as gn_apply_kernel (tdx_groupnorm.hip): n = fma(h, rstd gamma, beta - mean rstd gamma).
struct Conv1GnAdd {
    const float* stats;   // [B][G][2] (mean, rstd), or nullptr: plain addend
    const float* gamma;   // [Cout]
    const float* beta;    // [Cout]
    int G;
    int64_t V;            // voxels per sample (rows = B V)
};

// HF: format of the 16-bit tensors (H16<HF>: bf16 or fp16 words behind the bf16-typed pointers); the fp32 weights are
// rounded to it while staging
template <int NT, bool HF>
__global__ void __launch_bounds__(256)
conv1_mfma_fwd_kernel(const bf16* __restrict__ x1, int C1, const bf16* __restrict__ x2, int C2,
                      const float* __restrict__ w, int ldw, const float* __restrict__ bias,
                      const bf16* __restrict__ add, bf16* __restrict__ y, int64_t rows, int Cout, Conv1GnAdd gn) {
    typedef H16<HF> H;
    typedef typename H::T HT;
    constexpr int BN = 32 * NT;
    // LDS: staging (x slice 16 KB + w slice BN*64 B) and the output tile [256][64] bf16 (32 KB)
    // share one region
    __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
    unsigned char* sA = smem;
    unsigned char* sB = smem + C1M_ROWS * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * C1M_ROWS;
    const int n0 = blockIdx.y * BN;
    const int Cin = C1 + C2;

    f32x16 acc[NT][2];  // [n tile][m tile]: D[row = channel][col = voxel]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    for (int k0 = 0; k0 < Cin; k0 += C1M_KC) {
        const bf16* xs; int Cs, kk;
        if (k0 < C1) { xs = x1; Cs = C1; kk = k0; } else { xs = x2; Cs = C2; kk = k0 - C1; }
        uint4 areg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = tid + i * 256;          // 1024 pieces: row = pc >> 2, chunk = pc & 3
            const int64_t rr = row0 + (pc >> 2);
            areg[i] = make_uint4(0, 0, 0, 0);
            if (rr < rows) areg[i] = *reinterpret_cast<const uint4*>(xs + rr * Cs + kk + (pc & 3) * 8);
        }
        // weight slice: BN rows (n) x 4 chunks of 8 k; thread -> (n = p % BN, chunk = p / BN)
        uint4 breg[(BN * 4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (BN * 4 + 255) / 256; ++i) {
            const int pc = tid + i * 256;
            breg[i] = make_uint4(0, 0, 0, 0);
            if (pc < BN * 4) {
                const int n = pc % BN, c = pc / BN;
                const float* wp = w + (size_t)(k0 + c * 8) * ldw + n0 + n;
                unsigned u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    u[j] = H::pack2(wp[(size_t)(2 * j) * ldw], wp[(size_t)(2 * j + 1) * ldw]);
                breg[i] = make_uint4(u[0], u[1], u[2], u[3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = tid + i * 256;
            *reinterpret_cast<uint4*>(sA + sw64(pc >> 2, pc & 3)) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < (BN * 4 + 255) / 256; ++i) {
            const int pc = tid + i * 256;
            if (pc < BN * 4) *reinterpret_cast<uint4*>(sB + sw64(pc % BN, pc / BN)) = breg[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[2], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                xf[mt] = *reinterpret_cast<const bf16x8*>(sA + sw64(wave * 64 + mt * 32 + r, 2 * ks + hh));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const bf16x8*>(sB + sw64(nt * 32 + r, 2 * ks + hh));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = H16<HF>::mfma(wf[nt], xf[mt], acc[nt][mt]);
        }
    }

    // ---- epilogue through LDS, 64 output channels (2 n-tiles) per pass
    // lane owns voxel row (wave*64 + mt*32 + r) and channels nt*32 + 8 j + 4 hh + (0..3)
#pragma unroll
    for (int np = 0; np < NT; np += 2) {
        constexpr int PW = (NT >= 2) ? 64 : 32;  // channels per pass
        __syncthreads();
#pragma unroll
        for (int nn = 0; nn < (NT >= 2 ? 2 : 1); ++nn) {
            const int nt = np + nn;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int vr = wave * 64 + mt * 32 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = nn * 32 + 8 * j + 4 * hh;  // channel within the pass
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = bias[n0 + np * 32 + ch + e];
                    }
                    const unsigned lo = H::pack2(acc[nt][mt][4 * j] + bv[0], acc[nt][mt][4 * j + 1] + bv[1]);
                    const unsigned hi = H::pack2(acc[nt][mt][4 * j + 2] + bv[2], acc[nt][mt][4 * j + 3] + bv[3]);
                    int a;
                    if (PW == 64) a = sw128(vr, ch >> 3) + (ch & 7) * 2;
                    else a = sw64(vr, ch >> 3) + (ch & 7) * 2;
                    *reinterpret_cast<uint2*>(smem + a) = make_uint2(lo, hi);
                }
            }
        }
        __syncthreads();
        constexpr int CHUNKS = PW / 8;  // 16-B chunks per row
        constexpr int NI = (C1M_ROWS * CHUNKS) / 256;  // row chunks per thread; its chunk column c is the same in all of them
        const int c = tid % CHUNKS;
        const int cb = n0 + np * 32 + c * 8;  // first of the thread's 8 output channels
        if (add) {
            // all addend loads of the thread go out before the first is used (they were one memory round trip per
            // chunk), and the GroupNorm coefficients of its 8 channels are formed once per sample instead of per chunk
            // (a 64-bit division, 8 statistics loads and 16 parameter loads per 16 B: the fused tail ran at half the
            // bandwidth of the plain addend form, profiles/r11bf16_kernel_stats.csv)
            Raw8<HT> braw[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int64_t rr = row0 + (tid + i * 256) / CHUNKS;
                braw[i].load(reinterpret_cast<const HT*>(add + (rr < rows ? rr : rows - 1) * Cout + cb));
            }
            float ka[8], c0[8];
            int64_t s_end = -1;  // rows below s_end (and not below the sample's first row) use ka / c0 as they are
            auto coef = [&](int64_t rr) {
                const int bs = (int)(rr / gn.V), cpg = Cout / gn.G;
                s_end = ((int64_t)bs + 1) * gn.V;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float2 st = *reinterpret_cast<const float2*>(gn.stats + ((size_t)bs * gn.G + (cb + e) / cpg) * 2);
                    const float gam = gn.gamma[cb + e];
                    ka[e] = st.y * gam * 1.0f;
                    c0[e] = gn.beta[cb + e] - st.x * st.y * gam;
                }
            };
            // (the last row tile's rows >= `rows` are never stored: clamp, so that no thread forms coefficients from a
            // sample index == B, one past the (B, G, 2) statistics)
            if (gn.stats != nullptr) coef(row0 + tid / CHUNKS < rows ? row0 + tid / CHUNKS : rows - 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int vr = (tid + i * 256) / CHUNKS;
                const int64_t rr = row0 + vr;
                if (rr >= rows) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + (PW == 64 ? sw128(vr, c) : sw64(vr, c)));
                Vec8<HT> a, b = braw[i].get();
                a.load(reinterpret_cast<const HT*>(&v));
                if (gn.stats != nullptr) {
                    if (rr >= s_end) coef(rr);  // the workgroup's rows run into the next sample (rows ascend with i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) b.v[e] = silu_f(__builtin_fmaf(b.v[e], ka[e], c0[e]));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) a.v[e] += b.v[e];
                a.store(reinterpret_cast<HT*>(y + rr * Cout + cb));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int vr = (tid + i * 256) / CHUNKS;
                const int64_t rr = row0 + vr;
                if (rr < rows)
                    *reinterpret_cast<uint4*>(y + rr * Cout + cb) =
                        *reinterpret_cast<const uint4*>(smem + (PW == 64 ? sw128(vr, c) : sw64(vr, c)));
            }
        }
    }
}

int conv1_mfma_fwd_launch(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                          const void* add, void* y, int64_t rows, int Cout, hipStream_t st, const float* gn_stats,
                          const float* gn_gamma, const float* gn_beta, int gn_groups, int64_t gn_voxels, bool hf) {
    const int NT = (Cout % 64 == 0) ? 2 : 1;  // NT = 4 needs 270 registers: 1 wave/SIMD, too few loads in flight
    dim3 grid(ceil_div(rows, C1M_ROWS), Cout / (32 * NT));
    const Conv1GnAdd gn = {gn_stats, gn_gamma, gn_beta, gn_groups, gn_voxels};
#define C1M_LAUNCH(NTV, HFV)                                                                                             \
    hipLaunchKernelGGL((conv1_mfma_fwd_kernel<NTV, HFV>), grid, dim3(256), 0, st, (const bf16*)x1, C1, (const bf16*)x2, \
                       C2, w, ldw, bias, (const bf16*)add, (bf16*)y, rows, Cout, gn)
    if (NT == 2) { if (hf) C1M_LAUNCH(2, true); else C1M_LAUNCH(2, false); }
    else { if (hf) C1M_LAUNCH(1, true); else C1M_LAUNCH(1, false); }
#undef C1M_LAUNCH
    return tdx_launch_status();
}

// ------------------------------------------------------------------ weight gradient ------
__device__ __forceinline__ bf16x8 tr_frag8(const unsigned char* lo, const unsigned char* hi) {
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi));
    s16x8 rr = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, rr);
}

bool conv1_wgrad_mfma_supported(int Cin, int Cout) { return (Cin % 32) == 0 && (Cout % 32) == 0; }

#define C1W_PLANE (C1M_ROWS * 64)  // one [256][32] bf16 plane

// TR: the result is stored as dw[co][ci] (nn.Conv3d's own layout, row stride ldw): the MFMA is issued with its
// operands swapped, so that a lane owns one ci and the 32 lanes of a half-wave still write one 128-B run.
template <int MT, int NT, bool TR, bool HF>  // tile = (32 MT) ci x (32 NT) co; HF: operand format
__global__ void __launch_bounds__(256)
conv1_wgrad_mfma_kernel(const bf16* __restrict__ x, int Cin, const bf16* __restrict__ dy, int Cout,
                        float* __restrict__ dw, int ldw, float* __restrict__ dbias, int64_t rows, int nsplit,
                        int n_ci_tiles, int64_t split_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(MT + NT) * C1W_PLANE];
    unsigned char* sX = smem;
    unsigned char* sG = smem + MT * C1W_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / nsplit, split = blockIdx.x - tile * nsplit;
    const int ci0 = (tile % n_ci_tiles) * 32 * MT, co0 = (tile / n_ci_tiles) * 32 * NT;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int col_off = (16 * (g & 1) + 4 * p) * 2;
    const int kh = g >> 1;
    dw += (int64_t)split * split_stride;  // TDX_DETERMINISTIC: split k merges into its own zeroed slab
    if (dbias) dbias += (int64_t)split * split_stride;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    const bool do_bias = dbias != nullptr && ci0 == 0;
    const int64_t nchunks = (rows + C1M_ROWS - 1) / C1M_ROWS;

    // software pipeline: the rows of chunk ch + nsplit are in flight (registers) while chunk ch is
    // reduced out of LDS.  A thread always stages the same 8 output channels (chunk tid & 3 of plane
    // (tid >> 2) % NT), so the bias gradient is accumulated from its registers, not re-read from LDS.
    uint4 xr[MT * 4], gr[NT * 4];
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    auto load_chunk = [&](int64_t ch) {
        const int64_t row0 = ch * C1M_ROWS;
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const int pc = tid + i * 256;  // (row, plane, chunk): chunk fastest
            const int c = pc & 3, pl = (pc >> 2) % MT, rr = pc / (4 * MT);
            xr[i] = make_uint4(0, 0, 0, 0);
            if (row0 + rr < rows) xr[i] = *reinterpret_cast<const uint4*>(x + (row0 + rr) * Cin + ci0 + pl * 32 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const int pc = tid + i * 256;
            const int c = pc & 3, pl = (pc >> 2) % NT, rr = pc / (4 * NT);
            gr[i] = make_uint4(0, 0, 0, 0);
            if (row0 + rr < rows) gr[i] = *reinterpret_cast<const uint4*>(dy + (row0 + rr) * Cout + co0 + pl * 32 + c * 8);
        }
    };
    if (split < nchunks) load_chunk(split);
    for (int64_t ch = split; ch < nchunks; ch += nsplit) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT * 4; ++i) {
            const int pc = tid + i * 256;
            const int c = pc & 3, pl = (pc >> 2) % MT, rr = pc / (4 * MT);
            *reinterpret_cast<uint4*>(sX + pl * C1W_PLANE + rr * 64 + c * 16) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < NT * 4; ++i) {
            const int pc = tid + i * 256;
            const int c = pc & 3, pl = (pc >> 2) % NT, rr = pc / (4 * NT);
            *reinterpret_cast<uint4*>(sG + pl * C1W_PLANE + rr * 64 + c * 16) = gr[i];
            if (do_bias) {
                const unsigned wds[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bs[2 * e] += H16<HF>::lo(wds[e]);
                    bs[2 * e + 1] += H16<HF>::hi(wds[e]);
                }
            }
        }
        __syncthreads();
        if (ch + nsplit < nchunks) load_chunk(ch + nsplit);
        // wave reduces rows [64 wave, 64 wave + 64): 4 K-steps of 16 rows
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = wave * 64 + 16 * s + 8 * kh + q;
            bf16x8 af[MT], bfv[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned char* ap = sX + m * C1W_PLANE + row * 64 + col_off;
                af[m] = tr_frag8(ap, ap + 4 * 64);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const unsigned char* bp = sG + n * C1W_PLANE + row * 64 + col_off;
                bfv[n] = tr_frag8(bp, bp + 4 * 64);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = TR ? H16<HF>::mfma(bfv[n], af[m], acc[m][n])
                                   : H16<HF>::mfma(af[m], bfv[n], acc[m][n]);
        }
    }
    // ---- merge.  Each wave holds a private (32 MT) x (32 NT) tile: the four are summed through LDS first (fixed order), so
    // that a workgroup issues one tile's worth of atomics instead of four -- on the deep U-Net levels, where a workgroup
    // walks one or two row chunks, the atomics WERE the kernel (32 MB per launch at the chip's 1.3 TB/s of atomic adds:
    // 45-50 us per launch whatever the batch, profiles/r11_batch_scaling.txt).
    const int r = lane & 31, hh = lane >> 5;
    constexpr int TILE_F = (32 * MT) * (32 * NT);              // floats per tile; 4 tiles = (MT NT) x 16 KB <= the staging LDS
    static_assert(4 * TILE_F * 4 <= (MT + NT) * C1W_PLANE, "the four wave tiles fit the staging buffers");
    __syncthreads();                                            // everybody is done reading the last chunk
    float* tiles = reinterpret_cast<float*>(smem);              // [wave][row of the result][col]
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int q = (i & 3) + 8 * (i >> 2) + 4 * hh;  // the MFMA tile row this accumulator register holds
                // result row / column: TR stores dw[co][ci] (rows = co), else dw[ci][co]
                const int rr = TR ? n * 32 + q : m * 32 + q, cc = TR ? m * 32 + r : n * 32 + r;
                tiles[(wave * (TR ? 32 * NT : 32 * MT) + rr) * (TR ? 32 * MT : 32 * NT) + cc] = acc[m][n][i];
            }
    __syncthreads();
    {
        constexpr int ROWS = TR ? 32 * NT : 32 * MT, COLS = TR ? 32 * MT : 32 * NT;
        const int row_base = TR ? co0 : ci0, col_base = TR ? ci0 : co0;
        for (int e = tid; e < ROWS * COLS; e += 256) {
            const float t = (tiles[e] + tiles[ROWS * COLS + e]) + (tiles[2 * ROWS * COLS + e] + tiles[3 * ROWS * COLS + e]);
            atomicAdd(&dw[(size_t)(row_base + e / COLS) * ldw + col_base + e % COLS], t);
        }
    }
    if (do_bias) {
        // threads with equal (tid & 3, (tid >> 2) % NT) hold partial sums of the same 8 channels
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // [256][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 32 * NT) {
            const int pl = tid >> 5, c = (tid & 31) >> 3, e = tid & 7;
            float t = 0.f;
            for (int k = 0; k < 256; ++k)
                if ((k & 3) == c && ((k >> 2) % NT) == pl) t += red[k * 8 + e];
            atomicAdd(&dbias[co0 + tid], t);
        }
    }
}

int conv1_wgrad_mfma_launch(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                            int64_t rows, bool transposed, hipStream_t st, bool hf, int max_split, int64_t split_stride,
                            int* nsplit_out, bool plan_only) {
    const int MT = (Cin % 64 == 0) ? 2 : 1, NT = (Cout % 64 == 0) ? 2 : 1;
    const int n_ci = Cin / (32 * MT), n_co = Cout / (32 * NT);
    const int ntiles = n_ci * n_co;
    const int64_t nchunks = (rows + C1M_ROWS - 1) / C1M_ROWS;
    // one full wave of resident workgroups (LDS-limited occupancy), so that no partial second round trails
    const int per_cu = 160 / ((MT + NT) * 16);
    int nsplit = (per_cu * 256 + ntiles - 1) / ntiles;
    if (nsplit > nchunks) nsplit = (int)nchunks;
    if (nsplit < 1) nsplit = 1;
    if (max_split > 0 && nsplit > max_split) nsplit = max_split;
    if (nsplit_out) *nsplit_out = nsplit;
    if (plan_only) return TDX_OK;
    dim3 grid((unsigned)(ntiles * nsplit));
#define C1W_LAUNCH(M, N, T)                                                                                                 \
    do {                                                                                                                    \
        if (hf)                                                                                                            \
            hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<M, N, T, true>), grid, dim3(256), 0, st, (const bf16*)x, Cin,       \
                               (const bf16*)dy, Cout, dw, ldw, dbias, rows, nsplit, n_ci, split_stride);                    \
        else                                                                                                                \
            hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<M, N, T, false>), grid, dim3(256), 0, st, (const bf16*)x, Cin,      \
                               (const bf16*)dy, Cout, dw, ldw, dbias, rows, nsplit, n_ci, split_stride);                    \
    } while (0)
#define C1W_PICK(T)                           \
    if (MT == 2 && NT == 2) C1W_LAUNCH(2, 2, T); \
    else if (MT == 2) C1W_LAUNCH(2, 1, T);       \
    else if (NT == 2) C1W_LAUNCH(1, 2, T);       \
    else C1W_LAUNCH(1, 1, T)
    if (transposed) { C1W_PICK(true); } else { C1W_PICK(false); }
#undef C1W_PICK
#undef C1W_LAUNCH
    return tdx_launch_status();
}
