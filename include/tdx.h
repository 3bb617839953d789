/*
 * tdx.h -- C ABI of the MI355X (gfx950) turbdiff denoising-diffusion hot path.
 *
 * The upstream reference (martenlienen/generative-turbulence) has no FFI layer: its hot
 * path is stock PyTorch ops called from turbdiff/models/ddpm.py.  Each entry point below
 * replaces the ATen call(s) at the cited reference line; the host-side mirror
 * (generative-turbulence_amd/turbdiff_amd) binds them with ctypes and INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Activations are NDHWC ("voxel-major"): element (b, x, y, z, c) of a tensor with
 *     grid (X, Y, Z) and C channels lives at ((((b*X + x)*Y + y)*Z + z)*C + c).  The
 *     reference's NCDHW tensors (B, C, X, Y, Z) are converted at the model boundary by
 *     tdx_ncv_to_nvc / tdx_nvc_to_ncv (or fused into encode/decode).
 *   - dtype: TDX_F32 = 0 (float), TDX_BF16 = 1 (bfloat16 storage, fp32 accumulation), TDX_F16 = 3 (IEEE half storage,
 *     fp32 accumulation: the same kernels as TDX_BF16 with v_mfma_f32_32x32x16_f16 -- 11 significand bits, the precision of
 *     the reference's TF32 GPU convs (train.py:144-156), at the bfloat16 kernels' speed).
 *     Parameters, statistics, gradients of parameters and schedule tables are always f32.
 *   - Every pointer is a DEVICE pointer.  No entry point allocates, frees or synchronises;
 *     work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     so a caller may capture any sequence of calls into a hipGraph.
 *   - Workspaces are caller-provided; sizes come from the *_workspace_bytes queries.
 *   - Return value: 0 on success, negative TDX_E* on bad arguments, positive hipError_t if
 *     a launch failed.  No exceptions cross the ABI.
 */
#ifndef TDX_H
#define TDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDX_F32 0
#define TDX_BF16 1

#define TDX_OK 0
#define TDX_EINVAL (-1)   /* bad argument (null pointer, non-positive size)            */
#define TDX_ESHAPE (-2)   /* shape not supported by this kernel (e.g. C % 8 != 0)       */
#define TDX_EDTYPE (-3)   /* dtype not supported by this entry point                    */

/* conv3 implementation selector (tdx_conv3_*'s `impl` argument) */
#define TDX_CONV_AUTO 0    /* MFMA implicit GEMM when dtype/shape allow, else direct     */
#define TDX_CONV_DIRECT 1  /* vector-ALU reference kernels (any dtype)                    */
#define TDX_CONV_MFMA 2    /* MFMA implicit GEMM; TDX_ESHAPE if unsupported               */
#define TDX_CONV_SPLIT 3   /* fp32 tensors only: split-precision MFMA -- every operand as bf16 hi + lo,
                              x*w ~= xh*wh + xl*wh + xh*wl with fp32 accumulation (~4e-6 rel-L2 per layer
                              instead of ~3e-7; inside the 1e-4 gate, 2-3x faster than the fp32 MFMA).
                              Operands must come from tdx_conv3_pack_weight(dtype = TDX_F32_SPLIT); shapes the
                              split kernel does not cover run as TDX_CONV_AUTO                */
/* dtype code accepted by tdx_conv3_pack_weight only: fp32 weights packed for TDX_CONV_SPLIT */
#define TDX_F32_SPLIT 2
/* fp16 tensors: every entry point that takes TDX_BF16 takes TDX_F16 (round 6; BASELINE configs[4] names fp16 MFMA QK^T / AV) */
#define TDX_F16 3
/* OR-able into `impl` of tdx_conv3_fwd_gn and tdx_conv3_bwd_weight: the caller guarantees that the
 * workspace is all-zero on entry; the call skips its memsets and, as always, leaves the workspace
 * all-zero on exit (the kernels that read the accumulators clear them).  Lets a host keep one
 * persistent zeroed workspace per shape instead of paying ~90 memset launches per training step. */
#define TDX_WS_CLEAN 0x100

int tdx_version(void);
/* Scratch arena for kernels that need transient device workspace their entry point has no argument for: the K-split
 * slabs of the small-grid 3x3x3 conv (ddpm.py:164 on the 24x8x6 and 12x4x3 levels: several workgroups share an output
 * tile's K range, fp32 partial tiles go through the arena, a reduce kernel finishes them).  The library never
 * allocates: the caller hands over `bytes` (>= 64; 96 MiB covers the shipped model at B <= 8) of device memory
 * whose first 64 bytes are zero, keeps it alive and otherwise untouched, and orders all tdx_conv3_* launches on one
 * stream (or re-registers per stream).  The first 64 bytes are the zero source of the LDS-DMA kernels (ring conv data
 * gradient, producer / consumer weight gradient): a stream that launches only those may register a 64-byte arena; a
 * layer whose slabs do not fit the registered bytes runs on the brick kernels, as all do without an arena (ptr = NULL). */
int tdx_set_scratch(void* ptr, size_t bytes);
/* name of the gfx target the library was built for ("gfx950") */
const char* tdx_arch(void);

/* ------------------------------------------------------------------ layout ------------- */
/* (B, C, V) -> (B, V, C) and back; replaces the implicit NCDHW layout of every op in
 * ddpm.py.  dtype_in/dtype_out may differ (f32 <-> bf16 / fp16 cast fused). */
int tdx_ncv_to_nvc(const void* x, void* y, int B, int C, int64_t V, int dtype_in, int dtype_out, void* stream);
int tdx_nvc_to_ncv(const void* x, void* y, int B, int C, int64_t V, int dtype_in, int dtype_out, void* stream);
/* plain cast of n elements */
int tdx_cast(const void* x, void* y, int64_t n, int dtype_in, int dtype_out, void* stream);

/* ------------------------------------------------------------------ conv 3x3x3 --------- */
/* nn.Conv3d(k=3, padding=1, padding_mode="replicate"), ddpm.py:164.
 *
 * Weight packing: w (Cout, Cin, 3, 3, 3) f32 as stored in the reference's state_dict ->
 *   wf  forward operand          (K = Cin,  N = Cout, tap = (dx+1)*9 + (dy+1)*3 + (dz+1))
 *   wb  data-gradient operand    (K = Cout, N = Cin, taps flipped, weights transposed)
 * each 27*Cin*Cout elements of `dtype`; either may be NULL.  The element order is an
 * implementation detail shared by pack and the consumers, a function of (dtype, K, N) only:
 * [K/16][27][N][16] for the bf16 / fp16 MFMA kernels (TDX_BF16 / TDX_F16: K % 16 == 0, N % 32 == 0), [K/8][27][N][8] for the fp32 MFMA
 * kernels (K % 8 == 0, N % 32 == 0), else [27][K][N]; dtype = TDX_F32_SPLIT: two bf16 images (hi, lo), each
 * [K/8][27][N][8], in the fp32 operand's buffer where K % 16 == 0 and N % 32 == 0, else the fp32 layouts. */
int tdx_conv3_pack_weight(const float* w, void* wf, void* wb, int Cin, int Cout, int dtype, void* stream);
/* The same for n weights (`jobs`: HOST array; wf and wb both required): the weights whose two operands use the
 * MFMA layouts are packed by one launch per 32 jobs -- after an optimiser step a training step re-packs all its
 * 3x3x3 weights, and 21 launches of ~12 us were what that cost. */
typedef struct {
    const float* w;
    void* wf;
    void* wb;
    int Cin, Cout;
} TdxPackJob;
int tdx_conv3_pack_weights(const TdxPackJob* jobs, int n, int dtype, void* stream);
/* dst[c][r] = src[r][c] (f32) for n matrices in one launch (`jobs`: HOST array): the [Cin][Cout] operands of
 * tdx_conv1_fwd from nn.Conv3d(k=1) weights (Cout, Cin). */
typedef struct {
    const float* src;
    float* dst;
    int rows, cols; /* of src */
} TdxTransposeJob;
int tdx_transpose_many(const TdxTransposeJob* jobs, int n, void* stream);

/* y[b,v,:] = bias + sum_tap sum_ci x[b, clamp(v+tap), ci] * wf[tap][ci][:]
 * The input may be the channel concatenation of two tensors (x1: C1 channels, x2: C2
 * channels, C2 = 0 and x2 = NULL for a single input) -- replaces torch.cat (ddpm.py:370).
 * bias may be NULL. */
int tdx_conv3_fwd(const void* x1, int C1, const void* x2, int C2, const void* wf, const float* bias, void* y,
                  int B, int X, int Y, int Z, int Cout, int dtype, int impl, void* stream);

/* Which 16-bit (bf16 / fp16) matrix-core kernel serves this shape (same call; the kernels form the same products and sum them in fp32
 * in different orders, so their bf16 results agree up to ~1 ulp on a few % of the elements, not bit for bit.  Kernel
 * selection depends on the grid, on B -- a launch must fill the chip -- and on whether a scratch arena is bound
 * (tdx_set_scratch), so the last bit of a sample's output can change with the batch size it is computed in (B = 1
 * sampling vs B = 6 training) and with the arena setting; TDX_CONV3_RING=0 in the environment keeps every shape on the
 * brick kernels for batch-invariant results.  A host needs this query only for bookkeeping, e.g. bench.py's per-kernel
 * roofline): 1 = the persistent LDS-DMA ring kernel (tdx_conv3_ring.hip: grids
 * of whole 8x8x8 bricks that fill the chip, i.e. the two finest U-Net levels), 0 = the brick / small-grid kernels.
 * For a data gradient pass the layer's (Cout, 0, Cin) as (C1, C2, Cout). */
int tdx_conv3_uses_ring(int C1, int C2, int Cout, int B, int X, int Y, int Z);
/* The ring kernel's brick depth along z for that call: 8 (8 MT/2 x 8 x 8 bricks), 4 (round 6: 8 MT x 8 x 4 bricks, for grids
 * like level 2 of the benchmark grid, 48 x 16 x 12, whose z extent is a multiple of 4 only), or 0 = not a ring launch.
 * TDX_RING_Z4 in the environment (read per call): 0 = never 4-deep, 2 = 4-deep wherever legal (tests). */
int tdx_conv3_ring_brick_depth(int C1, int C2, int Cout, int B, int X, int Y, int Z);
/* Kernel family that tdx_conv3_fwd / tdx_conv3_fwd_gn run for a call with these arguments (same bookkeeping purpose):
 * vector-ALU, brick MFMA kernels (tdx_conv3_mfma*.hip), small-grid kernel (tdx_conv3_small.hip), ring kernel. */
#define TDX_KERNEL_DIRECT 0
#define TDX_KERNEL_BRICK 1
#define TDX_KERNEL_SMALL 2
#define TDX_KERNEL_RING 3
int tdx_conv3_fwd_kernel(int C1, int C2, int Cout, int B, int X, int Y, int Z, int dtype, int impl);

/* Same convolution, additionally producing the GroupNorm(G, Cout, eps) statistics of its own
 * output -- stats [B][G][2] = (mean, rstd), as tdx_gn_stats would -- from per-channel moments
 * accumulated in the conv epilogue (Block.conv -> Block.norm, ddpm.py:169-170), which saves the
 * separate read pass over y.  gn_workspace: tdx_gn_workspace_bytes(B, Cout). */
int tdx_conv3_fwd_gn(const void* x1, int C1, const void* x2, int C2, const void* wf, const float* bias, void* y,
                     float* stats, int G, float eps, void* gn_workspace, int B, int X, int Y, int Z, int Cout,
                     int dtype, int impl, void* stream);

/* Forward over a channel SUBSET of a wider tensor, continued from a partial result:
 *     y = conv3(x1[..., 0:C1], wf) + bias + init
 * x1 rows are ld1 elements apart (ld1 >= C1: the first C1 channels of a [.., ld1] tensor are read);
 * init is a bf16 tensor [B][X][Y][Z][Cout], or [X][Y][Z][Cout] shared by all samples when
 * init_shared != 0 (NULL: zeros); it is added to the bf16-rounded conv result in the store loop.  Used for the first conv of the
 * U-Net, whose input is cat(encode_x(x), encode_c_local(c).expand(B)) (ddpm.py:495-501): the
 * conditioning half of that conv is the same for every sample of a batch and for every reverse step
 * of a sampling run, so it is computed once and passed as `init`.  16-bit (bf16 / fp16) MFMA path only
 * (C1 % 16 == 0, Cout % 32 == 0).  stats/G/eps/gn_workspace as in tdx_conv3_fwd_gn, or stats = NULL. */
int tdx_conv3_fwd_partial(const void* x1, int C1, int ld1, const void* wf, const float* bias, const void* init,
                          int init_shared, void* y, float* stats, int G, float eps, void* gn_workspace, int B, int X,
                          int Y, int Z, int Cout, int dtype, int impl, void* stream);

/* Data gradient of the above: dx[b,u,:] = sum over (v,tap) with clamp(v+tap) == u of
 * wf[tap][:, :] dy[b,v,:]  (adjoint of the replicate-padded conv, halo folded back onto
 * the boundary).  The result has C1 + C2 channels and is split into dx1 / dx2 (dx2 may be
 * NULL when C2 == 0).  If `accumulate` != 0 the result is added to dx1/dx2 instead of
 * overwriting.  workspace: tdx_conv3_bwd_data_workspace_bytes().
 * Reproducibility: on the brick / ring kernels the boundary voxels' halo-shell terms are ADDED onto the stored main term
 * (tdx_conv3_shell.hip): a voxel on exactly one face gets one read-add-write, edge and corner voxels get up to 7 hardware
 * atomics (global_atomic_pk_add_bf16 / _f16 / global_atomic_add_f32) in whatever order the workgroups finish, each with its own
 * rounding in the tensor's dtype -- those voxels (1-3 % of the boundary shell) are not bit-reproducible from run to run,
 * in bf16 within 2^-8 relative per add.  TDX_SHELL_DETERMINISTIC=1 (environment, read per call) replaces the atomics by
 * a position buffer in `workspace` and a fixed-order fold: bit-identical results, one more small launch per call.
 * Everything else in the forward / data-gradient path sums in a fixed order (weight-gradient merges use fp32 atomics).  The small-grid kernels of the
 * deep levels and impl = TDX_CONV_DIRECT are deterministic throughout.
 * TDX_DETERMINISTIC=1 (environment, read per call; implies the shell's ordered route) makes EVERY parameter gradient of the training
 * step bit-reproducible from run to run: the fp32 atomic merges of tdx_conv3_bwd_weight (bias gradient; weight gradient of the
 * vector-ALU path and of many-split fp32-tensor launches), tdx_conv1_bwd_weight(_oc), tdx_encode_bwd and tdx_decode_bwd become
 * per-split partials added in a fixed order (csrc/tdx_ordered.hip; the last three take their slabs from the tdx_set_scratch arena
 * and return TDX_EINVAL / fall back to one split without one); the forward's f64 statistics merges are ordered too
 * (tdx_conv3_fwd_gn = conv + a statistics pass with per-block tables; the loss sums on an exact grid).  +1.5 ms on the 21-ms B = 6 step.  The reference itself sets no
 * determinism flag (grep: none in train.py / config/); under torch the counterpart would be torch.use_deterministic_algorithms. */
size_t tdx_conv3_bwd_data_workspace_bytes(int B, int X, int Y, int Z, int Cin, int dtype, int impl);
int tdx_conv3_bwd_data(const void* dy, const void* wb, void* dx1, int C1, void* dx2, int C2, int accumulate,
                       int B, int X, int Y, int Z, int Cout, int dtype, int impl, void* workspace, void* stream);

/* Same, with a fused addend: dx1 = adjoint[:, 0:C1] + add1, dx2 = adjoint[:, C1:] + add2 (either
 * addend may be NULL).  Used by the ResnetBlock backward, where the gradient arriving over the
 * residual path (ddpm.py:197) would otherwise cost a separate three-pass add. */
int tdx_conv3_bwd_data_add(const void* dy, const void* wb, void* dx1, int C1, void* dx2, int C2, const void* add1,
                           const void* add2, int B, int X, int Y, int Z, int Cout, int dtype, int impl,
                           void* workspace, void* stream);

/* Weight + bias gradient.  dw is written in the reference's parameter layout
 * (Cout, Cin, 3, 3, 3) f32, dbias (Cout) f32 (may be NULL).  Overwrites.
 * workspace: tdx_conv3_bwd_weight_workspace_bytes(Cin, Cout) = accumulators (27*Cin*Cout + Cout floats:
 * zeroed by the call unless impl has TDX_WS_CLEAN, and left all-zero on return) followed by scratch
 * slabs for launches with few K-splits (never needs zeroing). */
size_t tdx_conv3_bwd_weight_workspace_bytes(int Cin, int Cout, int impl);
int tdx_conv3_bwd_weight(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dw, float* dbias,
                         int B, int X, int Y, int Z, int Cout, int dtype, int impl, void* workspace, void* stream);

/* ------------------------------------------------------------------ conv 1x1x1 / linear - */
/* nn.Conv3d(k=1) (ddpm.py:188,292-293,433,436,459) and nn.Linear seen as a per-row GEMM:
 *   y[r, :] = bias + x1[r, :] @ w[0:C1, :] + x2[r, :] @ w[C1:C1+C2, :]   (+ add[r, :])
 * w is [Cin][ldw] f32 row-major with Cout <= ldw (the transposed reference weight).
 * x2/add/bias may be NULL. */
int tdx_conv1_fwd(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                  const void* add, void* y, int64_t rows, int Cout, int dtype, void* stream);
/* The tail of a ResnetBlock with a projected skip in one pass (ddpm.py:176 + 188 + 197):
 *   y[b, v, :] = silu(GroupNorm(h)[b, v, :]) + bias + x1[b, v, :] @ w[0:C1, :] + x2[b, v, :] @ w[C1:, :]
 * h: the block's second conv output (B, V, Cout), stats: its (B, groups, 2) mean / rstd from tdx_conv3_fwd_gn, gamma /
 * beta: the GroupNorm affine.  Same arithmetic as tdx_conv1_fwd(..., add = NULL) into a temporary followed by
 * tdx_gn_apply(h, ..., res = temporary, act = 1), without writing or re-reading the temporary.  bf16 / fp16 tensors on the
 * matrix-core kernel only: TDX_EDTYPE / TDX_ESHAPE otherwise (run the two calls). */
int tdx_conv1_fwd_gn(const void* x1, int C1, const void* x2, int C2, const float* w, int ldw, const float* bias,
                     const void* h, const float* stats, const float* gamma, const float* beta, int groups, void* y,
                     int B, int64_t V, int Cout, int dtype, void* stream);
/* dw[ci][co] (+)= sum_r x[r, ci] dy[r, co]  (f32, [Cin][ldw]); dbias[co] = sum_r dy[r, co].
 * Overwrites (buffers are zeroed inside).  x has Cin channels (call twice for a concat). */
int tdx_conv1_bwd_weight(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                         int64_t rows, int dtype, void* stream);
/* The same sums stored as dw[co][ci] ([Cout][ldw] f32, ldw >= Cin) -- nn.Conv3d's own weight layout, so the result
 * IS the parameter gradient (for a concat call twice, the second time with dw + C1).  accumulate != 0: adds into
 * dw / dbias as they are (the caller zeroed them, e.g. as one allocation) instead of zeroing inside. */
int tdx_conv1_bwd_weight_oc(const void* x, int Cin, const void* dy, int Cout, float* dw, int ldw, float* dbias,
                            int accumulate, int64_t rows, int dtype, void* stream);

/* ------------------------------------------------------------------ model boundary ------ */
/* encode_x / encode_c_local (ddpm.py:433,436,495-501) fused with NCDHW -> NDHWC and the
 * channel concatenation:  y[b,v,0:D] = wx x[b,:,v] + bx,  y[b,v,D:2D] = wc c[:,v] + bc.
 * x (B, Fx, V) f32; c (Fc, V) f32 shared by the batch, or NULL (then y has D channels);
 * wx [D][Fx], wc [D][Fc] are the reference's (D, F, 1, 1, 1) parameters.  Fx = Fc = 4. */
int tdx_encode_fwd(const float* x, int Fx, const float* wx, const float* bx, const float* c, int Fc, const float* wc,
                   const float* bc, void* y, int B, int64_t V, int D, int dtype, void* stream);
/* parameter gradients (overwritten) and dc (Fc, V) = gradient w.r.t. the conditioning (may be
 * NULL).  x is data (the noised sample): no gradient is produced for it. */
int tdx_encode_bwd(const void* dy, const float* x, int Fx, const float* c, int Fc, const float* wc, float* dwx,
                   float* dbx, float* dwc, float* dbc, float* dc, int B, int64_t V, int D, int dtype, void* stream);
/* decode.1 (ddpm.py:459,505) fused with NDHWC -> NCDHW: y[b,f,v] = w[f,:] . h[b,v,:] + bias[f],
 * y (B, F, V) f32, F = 4. */
int tdx_decode_fwd(const void* h, const float* w, const float* bias, float* y, int B, int64_t V, int D, int F, int dtype,
                   void* stream);
int tdx_decode_bwd(const float* dy, const void* h, const float* w, void* dh, float* dw, float* db, int B, int64_t V, int D,
                   int F, int dtype, void* stream);

/* ------------------------------------------------------------------ GroupNorm+FiLM+SiLU - */
/* nn.GroupNorm(G, C, eps) -> [x*(scale+1)+shift] -> [SiLU] (-> [+ residual]),
 * ddpm.py:165-176, 197, 429, 472.
 * stats [B][G][2] f32 = (mean, rstd) with biased variance. */
size_t tdx_gn_workspace_bytes(int B, int C);
int tdx_gn_stats(const void* x, float* stats, int B, int64_t V, int C, int G, float eps, int dtype, void* workspace,
                 void* stream);
/* n = (x-mean)*rstd*gamma + beta; if scale: n = n*(1+scale[b,c]) + shift[b,c];
 * a = act ? silu(n) : n; y = a + (res ? res : 0).   scale/shift: [B][C] f32 or NULL. */
int tdx_gn_apply(const void* x, const float* stats, const float* gamma, const float* beta, const float* scale,
                 const float* shift, const void* res, void* y, int B, int64_t V, int C, int G, int act, int dtype,
                 void* stream);
/* y = silu(GroupNorm(x)) + encode(x_raw, c_raw): the tail of the U-Net's first ResnetBlock (ddpm.py:180-197), whose identity
 * skip is the encoder output cat(encode_x(x), encode_c_local(c_local)) of ddpm.py:495-501.  The skip is evaluated here from
 * the raw planes (x_raw (B, Fx, V) f32, c_raw (Fc, V) f32 shared by the batch or NULL; wx/wc [D][F], bx/bc [D]; C = 2 D, or D
 * without c_raw) instead of being written by tdx_encode_fwd and read back, and is rounded to `dtype` before the add: the
 * result is bit-identical to tdx_encode_fwd + tdx_gn_apply(res = its output, act = 1).  Fx = Fc = 4. */
int tdx_gn_apply_encoded(const void* x, const float* stats, const float* gamma, const float* beta, const float* x_raw, int Fx,
                         const float* wx, const float* bx, const float* c_raw, int Fc, const float* wc, const float* bc,
                         void* y, int B, int64_t V, int D, int G, int dtype, void* stream);
/* out = decode(silu(GroupNorm(x)) + res): the tail of the model's last ResnetBlock (decode[0], ddpm.py:429) fused with the
 * dim -> F 1x1 decoder and the NDHWC -> NCDHW layout change (decode[1], ddpm.py:505); inference only (the block output, which
 * the decoder's weight gradient would need, is never written).  w [F][C] f32, bias [F]; out (B, F, V) f32; F = 4, C / 8 a power
 * of two.  Bit-identical to tdx_gn_apply(res, act = 1) + tdx_decode_fwd. */
int tdx_gn_apply_decode(const void* x, const float* stats, const float* gamma, const float* beta, const void* res,
                        const float* w, const float* bias, float* out, int B, int64_t V, int C, int G, int F, int dtype,
                        void* stream);
/* Backward of tdx_gn_apply w.r.t. x, gamma, beta, scale, shift (the residual's gradient is
 * dy itself).  dgamma/dbeta [C], dscale/dshift [B][C] (NULL when no FiLM); overwritten.
 * workspace: tdx_gn_workspace_bytes(). */
int tdx_gn_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
               const float* scale, const float* shift, void* dx, float* dgamma, float* dbeta, float* dscale,
               float* dshift, int B, int64_t V, int C, int G, int act, int dtype, void* workspace, void* stream);

/* ------------------------------------------------------------------ trilinear resize --- */
/* F.interpolate(mode="trilinear", align_corners=True), ddpm.py:359-361, 367-369. */
int tdx_resize_fwd(const void* x, void* y, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C, int dtype,
                   void* stream);
/* adjoint: dx (input grid) = A^T dy (output grid) [+ add]; overwrites dx.  `add` (input grid, may be
 * NULL) is a second gradient of the resampled tensor -- in the U-Net the tensor that is down-sampled is
 * also the skip connection (ddpm.py:355-358), so its two gradients are summed here instead of in a
 * separate three-pass add. */
int tdx_resize_bwd(const void* dy, const void* add, void* dx, int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int C,
                   int dtype, void* stream);

/* ------------------------------------------------------------------ attention ---------- */
/* F.scaled_dot_product_attention on the to_qkv output (attention.py:9-15, ddpm.py:295-308).
 * qkv [B][N][3*H*D]: channels [0,HD) = q, [HD,2HD) = k, [2HD,3HD) = v, head-major inside
 * each third.  out [B][N][H*D].  lse [B][H][N] f32 (log-sum-exp of the scaled scores).
 * Long sequences (bf16 / fp16 tensors, N >= 128) run the matrix-core flash kernel; when the launch has more 256-query
 * blocks than the chip holds at once and not a whole number of rounds (BASELINE configs[4]: 1152 blocks on 512 slots), a
 * persistent stream-K schedule splits the (query block, key tile) space evenly over 512 workgroups and parks the partial
 * (O, m, l) of split blocks in the scratch arena (tdx_set_scratch, 35.7 MB behind its zero block; without one, or with
 * TDX_ATTN_STREAMK=0, one workgroup per block). */
int tdx_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, int D, int dtype, void* stream);
int tdx_attn_bwd(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, int B, int N, int H,
                 int D, int dtype, void* workspace, void* stream);
size_t tdx_attn_bwd_workspace_bytes(int B, int N, int H, int D);

/* ------------------------------------------------------------------ DDPM arithmetic ---- */
/* All tensors here are the reference's NCDHW (B, F, V) f32 tensors; t is int64 on device:
 * per-sample (t_stride = 1) or one shared scalar (t_stride = 0).  mask is a dense uint8
 * [V] with 1 = in-domain cell (the reference's flat `cell_idx` list, utils.py:22-28). */

/* mask[i] = 1 for i in cell_idx, else 0 */
int tdx_cell_mask(const int64_t* cell_idx, int64_t n_cells, uint8_t* mask, int64_t V, void* stream);

/* q_sample (ddpm.py:818-822): out = sqrt_ac[t] x0 + sqrt_1mac[t] noise; if keep_bcs
 * (noise_bcs = False, ddpm.py:837-838) cells outside the mask keep x0. */
int tdx_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, const int64_t* t,
                 int t_stride, const uint8_t* mask, int keep_bcs, float* out, int B, int F, int64_t V, void* stream);

/* One fused reverse step (ddpm.py:745-752 + 797-811 [+ 814 on the last step]):
 *   x0h  = recip[t] x_t - recipm1[t] eps;  [!noise_bcs: x0h = x_t outside mask]; [clip +-1]
 *   mean = coef1[t] x0h + coef2[t] x_t
 *   t == 0:  out = mean, then out = x_bcs outside the mask
 *   t  > 0:  out = mean + exp(log_betas[t]/2) z   (z masked to the domain if !noise_bcs)
 *            noise_bcs: outside the mask out = sqrt_ac[t] x_bcs + sqrt_1mac[t] z2
 * sched = 7 consecutive f32 tables of length T:
 *   recip, recipm1, coef1, coef2, log_betas, sqrt_ac, sqrt_1mac.
 * t is a device int64 scalar (graph replay updates it on device). z, z2 may be NULL when
 * unused. */
int tdx_p_sample_step(const float* x_t, const float* eps, const float* z, const float* z2, const float* x_bcs,
                      const uint8_t* mask, const float* sched, int T, const int64_t* t, int noise_bcs, int clip,
                      float* out, int B, int F, int64_t V, void* stream);

/* The same reverse step with its noise drawn inside the kernel (needs V % 4 == 0 and 16-byte aligned tensors): bit-identical
 * to tdx_randn_batched(z, ...); [tdx_randn_batched(z2, ...) if noise_bcs;] tdx_p_sample_step(...) with the same seed,
 * stream ids and offset -- lane i of trajectory b draws the normals those calls would have written to z[b][4i..4i+3] -- without
 * the two noise tensors ever touching HBM.  Afterwards *offset_dev += (noise_bcs ? 2 : 1) * F V / 4 and *t -= 1 (both on the
 * device, so a captured graph replays the whole ddpm.py:789-813 loop body). */
int tdx_p_sample_step_rng(const float* x_t, const float* eps, const float* x_bcs, const uint8_t* mask, const float* sched,
                          int T, int64_t* t, int noise_bcs, int clip, float* out, int B, int F, int64_t V, uint64_t seed,
                          const uint64_t* stream_ids, uint64_t* offset_dev, void* stream);

/* Masked loss (ddpm.py:845-852): loss = mean_b mean_{f, cells} err(eps_hat, noise),
 * err = squared (l1 = 0) or absolute (l1 = 1) error.  n_cells = number of mask ones.
 * Writes loss[0] and, if grad != NULL, d loss / d eps_hat (zero outside the mask). */
int tdx_masked_loss(const float* eps_hat, const float* noise, const uint8_t* mask, int64_t n_cells, int l1,
                    float* loss, float* grad, int B, int F, int64_t V, void* workspace, void* stream);
/* The same with n_cells read from device memory when the kernels run (int64 scalar): what a hipGraph-captured training step
 * (turbdiff_amd/training.py, GraphedTrainingStep) calls, so that geometries with different in-domain cell counts
 * (utils.py:13-19 `select_cells` over `cell_idx`, ddpm.py:848) replay one graph. */
int tdx_masked_loss_dyn(const float* eps_hat, const float* noise, const unsigned char* mask, const int64_t* n_cells_dev,
                        int l1, float* loss, float* grad, int B, int F, int64_t V, void* workspace, void* stream);
size_t tdx_masked_loss_workspace_bytes(void);

/* ------------------------------------------------------------------ data ingress / egress */
/* The callers either side of the path.  Samples are the HDF5 files' channels-last
 * cell lists: variable i is a (B, n_cells, d_i) f32 array; up to four variables fill the slots from the
 * left (unused: NULL, 0); F = sum d_i <= 32.  Dense tensors are the reference's (B, F, V) f32.
 *
 * tdx_grid_embed replaces OpenFOAMData.grid_embedding (ofles.py:220-240: zeros, one index_put per
 * variable, one per FIXED_VALUE boundary condition) followed by Normalization.normalize_grid
 * (normalization.py:19-23) with one pass that writes every element of x exactly once:
 *   x[b, f, v] = shift[f] + scale[f] * raw,   raw = ovr_val[ovr_of[v], f]   if bit f of ovr_mask[ovr_of[v]]
 *                                                 = sample_f[b, cell_of[v]]  else if cell_of[v] >= 0
 *                                                 = 0                        otherwise
 * cell_of (V int32: position in the cell list or -1) and the override table (ovr_of: V int32 row or -1,
 * may be NULL; ovr_val: rows x F; ovr_mask: rows) are per-geometry constants the host derives once from
 * cell_idx, boundaries and boundary conditions with the reference's write order.  shift = -mean/std and
 * scale = 1/std (both NULL: no normalisation); one fused multiply-add per element, as ATen's CPU addcmul. */
int tdx_grid_embed(const float* s0, int d0, const float* s1, int d1, const float* s2, int d2, const float* s3, int d3,
                   const int32_t* cell_of, const int32_t* ovr_of, const float* ovr_val, const uint32_t* ovr_mask,
                   const float* shift, const float* scale, float* x, int B, int64_t n_cells, int64_t V, void* stream);
/* tdx_grid_select replaces Normalization.denormalize_grid (normalization.py:25-29) + select_cells
 * (utils.py:14-15) + the channels-last split SampleStore.add_samples writes (metrics.py:52-58):
 *   o_i[b, c, j] = mean[f] + std[f] * x[b, f, cell_idx[c]]   (mean = std = NULL: plain gather) */
int tdx_grid_select(const float* x, const int64_t* cell_idx, const float* mean, const float* std, float* o0, int d0,
                    float* o1, int d1, float* o2, int d2, float* o3, int d3, int B, int64_t n_cells, int64_t V,
                    void* stream);
/* CellTypeLearnedEmbedding.forward (cell_type_embeddings.py:69-70): out[d, v] = table[types[v], d] with
 * types a uint8 [V] grid (values < n_types <= 8, D <= 16), and the table gradient
 * dtable[k, d] (+)= sum over voxels of type k of dC[d, v] (fp64 partial sums, fixed order). */
int tdx_cell_embed_fwd(const uint8_t* types, const float* table, float* out, int n_types, int D, int64_t V, void* stream);
size_t tdx_cell_embed_bwd_workspace_bytes(int n_types, int D);
int tdx_cell_embed_bwd(const uint8_t* types, const float* dC, float* dtable, int accumulate, int n_types, int D, int64_t V,
                       void* workspace, void* stream);

/* ------------------------------------------------------------------ sample metrics ------ */
/* TurbulentKineticEnergySpectrum.forward (turbdiff/models/metrics.py:289-316) around the FFT (section 8 f3).
 * tdx_tke_energy: tke[b, v] = 0.5 * sum_c u[b, c, v]^2 for u (B, 3, V) f32 (metrics.py:291).
 * tdx_tke_sphere: fft is the UNSHIFTED complex64 fftn of tke, (B, X, Y, Z) interleaved (re, im); p (N, 3) and
 * w (N) a quadrature rule on the unit sphere, k (K) radii:
 *   E[b, k] = 4 pi k^2 sum_n w[n] exp(interp3(log |fftshift(F)|^2, k p[n] + (X/2, Y/2, Z/2)))
 * with interp3 of metrics.py:220-268 (corners clamped into the grid, weights relative to the clamped lower
 * corner). */
int tdx_tke_energy(const float* u, float* tke, int B, int64_t V, void* stream);
int tdx_tke_sphere(const float* fft, const float* p, const float* w, const float* k, float* E, int B, int X, int Y, int Z,
                   int N, int K, void* stream);

/* Counter-based N(0,1) generator (Philox4x32-10 + Box-Muller), graph-replay safe: the
 * 64-bit offset is read from device memory and advanced by the kernel itself.
 * Replaces torch.randn_like (ddpm.py:777,801,810,835) inside captured sampling graphs.
 * Stream `stream_id` separates trajectories so results do not depend on how trajectories
 * are sharded over GPUs. */
int tdx_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t* offset_dev, void* stream);
/* Batched form: out is [B][n]; row b uses Philox stream stream_ids[b] (device array of B
 * uint64 trajectory ids) and every row the same offset, which is then advanced once. */
int tdx_randn_batched(float* out, int B, int64_t n, uint64_t seed, const uint64_t* stream_ids, uint64_t* offset_dev,
                      void* stream);

/* ---- a point inside a captured graph that the HOST can see -----------------------------------------
 * Stores *gen_dev * TDX_SIGNAL_STRIDE + k (k < TDX_SIGNAL_STRIDE) into *host_flag -- pinned host memory mapped into the
 * device's address space -- with a system-scope release, in stream order: when the host reads that value, everything
 * enqueued before the call on `stream` has completed and is visible to later launches on any stream.  Used by the
 * data-parallel captured training step (parallel.BucketedDataParallel.replay_launch): the reference's DDP (Lightning's
 * strategy around torch.nn.parallel.DistributedDataParallel, train.py) launches a bucket's all-reduce from an autograd hook;
 * a replayed hipGraph runs no hooks, and this ROCm runtime has no event-record graph nodes, so the host polls the word a
 * captured kernel writes behind each bucket's staging and launches the collective itself.  gen_dev: a device counter the
 * graph increments once per replay, so the values rise monotonically and the word never needs resetting. */
#define TDX_SIGNAL_STRIDE 64
int tdx_signal_host(uint32_t* host_flag, const uint32_t* gen_dev, uint32_t k, void* stream);

/* ---- optimiser tail of the training step ------------------------------------------------------
 * Replaces torch.nn.utils.clip_grad_norm_ (Lightning gradient_clip_val = 0.1, config/train.yaml:30-31)
 * and torch.optim.RAdam.step (models/diffusion.py:210-218; betas (0.9, 0.999), eps 1e-8, no weight
 * decay) over all parameter tensors at once.  `table`: device array of one TdxOptTensor per
 * parameter (fp32, contiguous; grad == NULL: the parameter is skipped).  The element range of every
 * tensor is cut into chunks of tdx_opt_chunk_elems(); chunk c covers tensor chunk_tensor[c] from
 * element chunk_off[c] (device arrays of nchunks entries, built once by the host). */
typedef struct {
    void* param;
    void* grad;
    void* exp_avg;
    void* exp_avg_sq;
    int64_t numel;
} TdxOptTensor;
int64_t tdx_opt_chunk_elems(void);
/* out[0] = L2 norm over all gradients, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
 * (1 when max_norm <= 0).  partial: nchunks floats of scratch.  No host synchronisation. */
int tdx_grad_norm(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                  float max_norm, float* partial, float* out, void* stream);
/* The same for LOSS-SCALED gradients (fp16 training: the backward pass ran on S x loss, S = 1 / inv_scale a power of two, so
 * that fp16 activation gradients stay representable; the reference has no such mode -- its GPU runs are TF32, train.py:144-156).
 * out: 4 floats -- out[0] = the TRUE gradient norm (stored norm x inv_scale), out[1] = inv_scale x clip coefficient (the
 * factor from a stored gradient to the clipped true one; 0 when the norm is not finite), out[2] = 1.0 when the norm is
 * inf / nan (fp16 overflow: tdx_radam_step_scaled then leaves parameters and moments untouched; the host halves S), else 0. */
int tdx_grad_norm_scaled(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                         float max_norm, float inv_scale, float* partial, float* out, void* stream);
/* tdx_radam_step with scaled_norm = the `out` of tdx_grad_norm_scaled: gradients x scaled_norm[1]; the whole step is a
 * no-op when scaled_norm[2] != 0. */
int tdx_radam_step_scaled(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                          const float* scaled_norm, int64_t step, float lr, float beta1, float beta2, float eps,
                          int write_grad, void* stream);
/* One RAdam step number `step` (1-based) with gradients scaled by clip[1] (clip: the `out` of
 * tdx_grad_norm, or NULL for no clipping); write_grad != 0 also stores the scaled gradients back,
 * as clip_grad_norm_ does. */
int tdx_radam_step(const TdxOptTensor* table, const int* chunk_tensor, const int64_t* chunk_off, int nchunks,
                   const float* clip, int64_t step, float lr, float beta1, float beta2, float eps, int write_grad,
                   void* stream);

/* ---- gradient staging of the data-parallel step ---------------------------------------------------
 * Replaces the per-parameter copy of a gradient into its all-reduce bucket that DistributedDataParallel's autograd hooks
 * perform (the reference trains under Lightning's DDP strategy, train.py:144-156): dst[k] = src[k] * scale for every item
 * in ONE launch per TDX_STAGE_MAX_ITEMS items (fp32; src == NULL: dst <- 0; src == dst: scaled in place).  `items` is a
 * HOST array (its pointers are device pointers) and is not read after the call returns; any alignment (16-byte aligned
 * pairs take the vector path). */
#define TDX_STAGE_MAX_ITEMS 64
typedef struct {
    const void* src;
    void* dst;
    int64_t n;
} TdxStageItem;
int tdx_stage_scaled(const TdxStageItem* items, int n_items, float scale, void* stream);

/* ---- FiLM projections of all ResnetBlocks at once -------------------------------------------------
 * Replaces, for every block of the U-Net in one launch, nn.Linear(c_dim, 2 * dim_out) on the conditioning vector and
 * the chunk into (scale, shift) (reference models/ddpm.py:184,191-192), and in the backward its three gradients.
 * c: (B, T) f32.  Layer i: weight (2 C_i, T), bias (2 C_i) or NULL, out (2, B, C_i) f32 -- out[0] = scale,
 * out[1] = shift, the dense (B, C) operands of tdx_gn_apply / tdx_gn_bwd.  `layers` is a HOST array of n <=
 * TDX_FILM_MAX_LAYERS entries (its pointers are device pointers); batch limits: tdx_film_supported. */
#define TDX_FILM_MAX_LAYERS 32
typedef struct {
    const float* weight;
    const float* bias;
    float* out;
    int channels; /* C_i */
} TdxFilmLayer;
/* 1 if tdx_film_fwd and tdx_film_bwd take a batch of B conditioning vectors of T features (they keep the batch's
 * vectors in LDS: B*T*4 bytes forward, (B*T + 64*(B+1) + 64*(T+1))*4 backward, at most 160 KiB); otherwise a host
 * projects per block with its own GEMM (turbdiff_amd.ops.film_projections does), as ddpm.py:191-192 does. */
int tdx_film_supported(int B, int T);
int tdx_film_fwd(const float* c, int B, int T, const TdxFilmLayer* layers, int n, void* stream);
/* grad_out (2, B, C_i) -> grad_weight (2 C_i, T), grad_bias (2 C_i) or NULL; dc (B, T) = the sum over all layers of
 * grad_out_i^T-stacked @ weight_i (overwritten).  Deterministic (fixed summation order).  workspace:
 * tdx_film_bwd_workspace_bytes(B, T, channels, n) bytes of scratch. */
typedef struct {
    const float* weight;
    const float* grad_out;
    float* grad_weight;
    float* grad_bias;
    int channels;
} TdxFilmGrad;
size_t tdx_film_bwd_workspace_bytes(int B, int T, const int* channels, int n);
int tdx_film_bwd(const float* c, int B, int T, const TdxFilmGrad* layers, int n, float* dc, void* workspace, void* stream);

/* ------------------------------------------------------------------ baseline conv variants
 * The conv layers of the reference's regression baselines, NDHWC, off the benchmark path:
 *   nn.Conv3d(dim, dim, 3, dilation=d, padding=d, padding_mode="replicate")        dilresnet.py:28-35
 *   nn.Conv3d(cin, cout, k, stride=2, padding=(k-1)//2)                            tfnet.py:187-193
 *   nn.ConvTranspose3d(cin, cout, kernel_size=4, stride=2, padding=1)              tfnet.py:203-205
 * Weights are passed as w[tap][Cin][Cout] f32, tap = (t0*k + t1)*k + t2, Cin = channels of `in`, Cout = channels of
 * `out` (for a transposed layer and for data gradients the caller passes the appropriately transposed matrix).
 *
 * tdx_convg_apply, transposed == 0 (conv forward; data gradient of a transposed conv):
 *     out[b, o, :] = bias + sum_t in[b, src(o, t), :] @ w[t],   src = o*stride - pad + t*dilation, taken with
 *     replicate != 0 by clamping to the grid, else skipped when outside.
 * transposed != 0 (transposed-conv forward; data gradient of a zero-padded conv):
 *     out[b, i, :] = bias + sum_t in[b, (i + pad - t*dilation)/stride, :] @ w[t]   where divisible and in range.
 * The data gradient of a REPLICATE-padded conv (stride 1) is the transposed form evaluated on the padded grid
 * (out grid = E + 2 pad, pad argument 0 ... see turbdiff_amd/ops.py) followed by tdx_convg_fold_clamp, which adds every
 * padded position onto the voxel it clamps to.  Cin % 8 == Cout % 8 == 0.
 * Arithmetic: fp32 tensors run on the vector ALU with fp32 weights; bf16 tensors (TDX_BF16) run on the matrix cores
 * (tdx_convg_mfma.hip) -- the fp32 weights are rounded to bf16 on their way into LDS, products accumulate in fp32, the
 * result is stored as bf16; TDX_CONVG_MFMA=0 (environment, read per call) keeps bf16 tensors on the vector-ALU kernels. */
int tdx_convg_apply(const void* in, const float* w, const float* bias, void* out, int B, int Xi, int Yi, int Zi, int Cin,
                    int Xo, int Yo, int Zo, int Cout, int k, int stride, int dilation, int pad, int replicate,
                    int transposed, int dtype, void* stream);
int tdx_convg_fold_clamp(const void* dpad, void* dx, int B, int X, int Y, int Z, int pad, int C, int dtype, void* stream);
/* dw[tap][Cin][Cout] (+ dbias[Cout], may be NULL) of out = apply(in, w) in the non-transposed form; accumulates with
 * f32 atomics: dw / dbias must be zero on entry.  Cout <= 512. */
int tdx_convg_bwd_weight(const void* in, const void* dy, float* dw, float* dbias, int B, int Xi, int Yi, int Zi, int Cin,
                         int Xo, int Yo, int Zo, int Cout, int k, int stride, int dilation, int pad, int replicate,
                         int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TDX_H */
