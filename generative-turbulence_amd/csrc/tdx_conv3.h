// Internal interface between tdx_conv3_direct.hip (entry points, vector-ALU kernels) and
// tdx_conv3_mfma.hip (bf16 MFMA implicit-GEMM kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Conv3Geom {
    int B;
    int Xi, Yi, Zi;  // input grid
    int Xo, Yo, Zo;  // output grid
    int off;         // output voxel o reads input voxel o + off + e, e in {-1,0,1}^3
};

// Packed-operand layouts (element index of weight (tap, k, n); K = input channels of the
// conv being evaluated, N = its output channels):
//   generic : (tap*K + k)*N + n                                   [27][K][N]
//   mfma    : (((k/kc)*27 + tap)*N + n)*kc + (k%kc)               [K/kc][27][N][kc]
// with kc = 16 for bf16 iff conv3_mfma_supported(K, 0, N), kc = 8 for fp32 iff
// conv3_mfma_f32_supported(K, 0, N), else the generic layout (kc = 0) -- a function of (dtype, K, N)
// only, so that pack and consumers agree without carrying a flag through the ABI.
bool conv3_mfma_supported(int C1, int C2, int Cout);
bool conv3_mfma_f32_supported(int C1, int C2, int Cout);
static inline int conv3_layout_kc(int dtype, int K, int N) {
    if (dtype == 1 || dtype == 3) return conv3_mfma_supported(K, 0, N) ? 16 : 0;  // TDX_BF16, TDX_F16: one layout
    return conv3_mfma_f32_supported(K, 0, N) ? 8 : 0;
}
static inline bool conv3_uses_mfma_layout(int dtype, int K, int N) { return (dtype == 1 || dtype == 3) && conv3_mfma_supported(K, 0, N); }
// split-precision (bf16 hi + lo) MFMA forward / zero-padded data gradient for fp32 tensors (tdx_conv3_mfma_split.hip)
bool conv3_mfma_split_supported(int C1, int C2, int Cout);
int conv3_mfma_split_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                            const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc = nullptr,
                            void* d1 = nullptr, int D1 = 0, void* d2 = nullptr, const void* a1 = nullptr, const void* a2 = nullptr);
// fp32 MFMA forward / zero-padded data gradient (tdx_conv3_mfma_f32.hip)
int conv3_mfma_f32_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                          const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc = nullptr,
                          void* d1 = nullptr, int D1 = 0, void* d2 = nullptr, const void* a1 = nullptr, const void* a2 = nullptr);

// optional extras of the MFMA forward: input row strides (0: dense) and a tensor the accumulators
// start from ([B][V][Cout] bf16, or [V][Cout] shared by the batch)
struct Conv3Ext {
    int ld1, ld2;
    const void* init;
    bool init_shared;
};
int conv3_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                      const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc = nullptr,
                      void* d1 = nullptr, int D1 = 0, void* d2 = nullptr, const void* a1 = nullptr,
                      const void* a2 = nullptr, const Conv3Ext* ext = nullptr, const int* slabs_beyond = nullptr,
                      bool hf = false);
// hf (here and in the launchers below): the tensors and the packed operand are IEEE half instead of bfloat16 (TDX_F16):
// same kernels, same layouts, v_mfma_f32_32x32x16_f16 and half rounding of the results (H16<HF>, tdx_common.h)
// slabs_beyond = {mx, my, mz}: another kernel has computed the region [0, mx) x [0, my) x [0, mz) of the output; launch
// only the thin-brick kernel on the remainder slabs beyond it (1-2 voxels thick per axis).

// persistent LDS-DMA ring kernel (tdx_conv3_ring.hip; bf16): forward or main term of the data gradient on grids whose
// whole 8 x 8 x 8 bricks fill the chip and leave remainders of at most 2 voxels per axis (those go to the thin-brick kernel
// of tdx_conv3_mfma.hip in a second launch: the reference's 194 x 50 x 50 and its 97 x 25 x 25 level); TDX_ESHAPE = not such a case, take conv3_mfma_launch (results equal up to the
// fp32 summation order: ~1 bf16 ulp on a few % of the elements)
bool conv3_ring_supported(int C1, int C2, int Cout, int B, int X, int Y, int Z);
int conv3_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y, int B, int X,
                      int Y, int Z, int Cout, bool zero_pad, hipStream_t st, double* gn_acc = nullptr, void* d1 = nullptr,
                      int D1 = 0, void* d2 = nullptr, const void* a1 = nullptr, const void* a2 = nullptr, bool hf = false);

// small-grid conv (tdx_conv3_small.hip; bf16 tensors, or fp32 tensors with split-precision products when split): forward
// or data gradient (then x1 = dy, result split over out1 / out2 with addends, halo fold included); TDX_ESHAPE = not a
// small-grid case, take the brick kernels.  Needs the scratch arena.
int conv3_small_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* out1, int D1,
                       void* out2, const void* add1, const void* add2, int B, int X, int Y, int Z, int N, bool data_gradient,
                       bool split, hipStream_t st, bool hf = false);
bool conv3_small_applies(int C1, int C2, int B, int X, int Y, int Z, int N, bool data_gradient, bool split);
// packed-K weight gradient for small grids (tdx_conv3_wgrad_small.hip, bf16); same contract as conv3_wgrad_mfma_launch,
// TDX_ESHAPE = not a small-grid case
int conv3_wgrad_small_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B,
                             int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out,
                             bool hf = false);
// producer / consumer weight gradient, 64- and 32-wide output tiles (tdx_conv3_wgrad_ring.hip, bf16: 8 computing + 4 loader
// waves per workgroup); same contract as conv3_wgrad_mfma_launch, TDX_ESHAPE = not a case for it
int conv3_wgrad_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias, int B, int X,
                            int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs, int* nslab_out, bool hf = false);
// CUs the persistent one-workgroup-per-CU kernels (ring conv, producer / consumer weight gradient) may occupy:
// TDX_PERSISTENT_CUS in the environment (read per call), a multiple of 8 in [8, 256], default 256.  A data-parallel run
// sets it below 256 to leave CUs to RCCL's kernels.
int tdx_persistent_cus();
// TDX_DETERMINISTIC=1 in the environment (read per call): the fp32 atomic merges of the backward's small parameter gradients are
// replaced by per-split partials added in a fixed order (tdx_ordered.hip), the halo shell takes its ordered route
bool tdx_deterministic();
// dst[r * ld + c] (+)= sum_{k < nslab} slabs[k * stride + r * cols + c], k ascending
int ordered_sum_launch(const float* slabs, int nslab, int64_t stride, float* dst, int rows, int cols, int64_t ld, bool add,
                       hipStream_t st);
// dbias[c] = sum over the nvox rows of dy[v][c] (NDHWC, C % 8 == 0) in a fixed order; part: >= C floats of scratch (256 C used if there)
size_t bias_grad_ordered_scratch_floats(int C);
int bias_grad_ordered_launch(const void* dy, int64_t nvox, int C, int dtype, float* dbias, float* part, size_t part_floats,
                             hipStream_t st);
// the caller-provided scratch arena (tdx_set_scratch, include/tdx.h); nullptr if none
void* tdx_scratch_ptr();
size_t tdx_scratch_bytes();

bool conv3_wgrad_mfma_supported(int C1, int C2, int Cout);
bool conv3_wgrad_mfma_split_supported(int C1, int C2, int Cout);
int conv3_wgrad_mfma_split_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias,
                                  int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs,
                                  int* nslab_out);
// producer / consumer form of the split-precision weight gradient (tdx_conv3_wgrad_split_ring.hip: 8 computing + 4 loader
// waves, 2 x 8 x 8 bricks); same contract, TDX_ESHAPE = not a case for it
int conv3_wgrad_split_ring_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias,
                                  int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs, int max_slabs,
                                  int* nslab_out);
bool conv3_wgrad_mfma_f32_supported(int C1, int C2, int Cout);
int conv3_wgrad_mfma_f32_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp,
                                float* dbias, int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs,
                                int max_slabs, int* nslab_out);
// slabs / max_slabs: optional region of max_slabs x 27*Cin*Cout floats; when the launch uses at most
// max_slabs K-splits every split stores its partial tiles there (no atomics) and *nslab_out tells the
// caller how many slabs to add up (0: the result was accumulated into dwp)
int conv3_wgrad_mfma_launch(const void* x1, int C1, const void* x2, int C2, const void* dy, float* dwp, float* dbias,
                            int B, int X, int Y, int Z, int Cout, hipStream_t st, float* slabs = nullptr,
                            int max_slabs = 0, int* nslab_out = nullptr, bool hf = false);

// halo-shell term of the data gradient, added onto dx with atomics (tdx_conv3_shell.hip); mode 0 bf16, 1 fp32 MFMA,
// 2 split-precision, 3 fp16; wb = the packed data-gradient operand for (K -> N) in that mode's layout
// sbuf: conv3_shell_buffer_bytes() of scratch, used when TDX_SHELL_DETERMINISTIC=1 (positions stored, then folded in a
// fixed order by a second kernel, instead of atomics on edge / corner voxels); nullptr: always the atomics route
size_t conv3_shell_buffer_bytes(int B, int X, int Y, int Z, int N);
int conv3_shell_launch(const void* dy, const void* wb, void* d1, int D1, void* d2, int B, int X, int Y, int Z, int K, int N,
                       int mode, hipStream_t st, void* sbuf = nullptr);

int conv3_direct_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                        const Conv3Geom& g, int Cout, int dtype, bool zero_pad, hipStream_t st);

// tdx_groupnorm.hip: (mean, rstd) per (b, group) from per-channel f64 (sum, sumsq)
#define TDX_GN_REPLICAS 32  // == GN_REPLICAS in tdx_groupnorm.hip (sizes tdx_gn_workspace_bytes)
int gn_finalize_launch(double* acc, float* stats, int B, int C, int G, int64_t V, float eps, int replicas,
                       hipStream_t st);
// tdx_gn_stats with the TDX_WS_CLEAN promise passed through (tdx_groupnorm.hip)
int gn_stats_launch(const void* x, float* stats, int B, int64_t V, int C, int G, float eps, int dtype, void* workspace,
                    bool clean, hipStream_t stream);
