// Split-precision ("bf16x2") MFMA 3x3x3 convolution for fp32 tensors: forward and, zero-padded on the
// padded grid, the data gradient.  gfx950.
//
// Every fp32 operand is split into two bf16 terms, v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (16 significant bits together), and the product is formed from three bf16 MFMAs with fp32 accumulation,
//     x * w  ~=  x_hi * w_hi + x_lo * w_hi + x_hi * w_lo            (the dropped lo * lo term is ~2^-18 relative)
// so a K = 16 step costs 3 x 32 cycles of v_mfma_f32_32x32x16_bf16 instead of the 8 x 64 cycles of
// v_mfma_f32_32x32x2_f32 (tdx_conv3_mfma_f32.hip): 5.3x fewer matrix-core cycles for a result that differs
// from the exact fp32 convolution by ~4e-6 rel-L2 per layer (1.4e-5 end to end on the 2-level golden
// model), i.e. inside the 1e-4 parity gate with room to spare.  It is opt-in (TDX_CONV_SPLIT): the default
// fp32 mode keeps IEEE fp32 products.
//
// Structure: a 4 x 8 x 8 brick of output voxels (8 x 8 x 8 for 32-wide tiles on large grids) and BN = 32 NT output channels
// per workgroup; the halo'd brick of an 8-CHANNEL slice and the slice's weights of all taps in LDS, split while staging:
// activations are loaded as fp32 and written as a hi and a lo image (16-B entries = 8 bf16 channels of a voxel, z stride
// padded 10 -> 12); weights arrive pre-split from tdx_conv3_pack_weight(TDX_F32_SPLIT) ([2][K/8][27][N][8] bf16: a slice's
// tap row is contiguous).  An MFMA K step of 16 is 8 channels x 2 TAPS: lanes 0-31 of a fragment hold tap 2 j, lanes 32-63 tap 2 j + 1
// (14 steps, the 28th tap meets zero weight rows), the arrangement of tdx_conv3_ring.hip.  That halves the LDS image against
// the round-4 form (16-channel slices, 157 KB): 80.6 KB for NT = 2, so TWO workgroups share a CU and one's staging, barriers
// and epilogue run beside the other's MFMAs; <= 256 registers per lane (236; the per-piece source offsets are rebuilt from a
// lane constant + uniform terms every slice instead of being kept).  The fragments of step j + 1 are read into a second
// register set while step j issues its 6 NT MFMAs, and the next slice's global loads are in flight during the MFMA phase.
//
// Where its time goes (profiles/r12_split_staging_ablation.txt; 128 -> 128 channels at 96 x 32 x 24 x 6): the matrix pipe is
// busy 64 % of the cycles at 1.61 GHz (round 4's form: 51 % at 1.77; the chip lowers the clock as the pipe fills: with
// staging and epilogue removed, 90 % at 1.56 GHz is the ceiling).  What is left is the global LOADS of the staging, not its
// LDS stores or the split arithmetic: dropping the loads of the activations gains 8 % (32-wide tiles on 8 x 8 x 8 bricks:
// 18 %), those of the weights 15 % (11 %); dropping only the stores, nothing.  An 8-channel slice uses 32 B of every
// 128-B line of an NDHWC fp32 tensor, and every workgroup re-reads the slice's 57 KB of weights from L2.
#include "tdx_conv3_mfma_split_kernel.h"

bool conv3_mfma_split_supported(int C1, int C2, int Cout) {
    return C1 > 0 && (C1 % SP_KC) == 0 && (C2 % SP_KC) == 0 && (Cout % 32) == 0;
}

int conv3_mfma_split_go_2zmp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_2zmn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_2rmp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_2rmn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1zbp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1zbn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1rbp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1rbn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1zmp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1zmn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1rmp(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1rmn(SPLIT_GO_ARGS);
int conv3_mfma_split_go_1ztp(SPLIT_GO_ARGS);

int conv3_mfma_split_launch(const void* x1, int C1, const void* x2, int C2, const void* wp, const float* bias, void* y,
                            const Conv3Geom& g, int Cout, bool zero_pad, hipStream_t st, double* gn_acc, void* d1, int D1, void* d2,
                            const void* a1, const void* a2) {
    if ((int64_t)g.Xi * g.Yi * g.Zi * 2 >= (1ll << 31) || (int64_t)g.Xo * g.Yo * g.Zo >= (1ll << 31)) return TDX_ESHAPE;
    static const bool no_thin = getenv("TDX_CONV3_THIN") && atoi(getenv("TDX_CONV3_THIN")) == 0;  // A/B switch
    const int NT = (Cout % 64 == 0) ? 2 : 1;
    // 32-wide output tiles: 8 x 8 x 8 bricks (four M tiles per wave) where the grid is large enough to fill the chip
    const bool big = NT == 1 && (int64_t)g.B * ceil_div(g.Xo, 8) * ceil_div(g.Yo, 8) * ceil_div(g.Zo, 8) >= 1024;
    BrickRegions main, thin;
    brick_plan(g, zero_pad, !no_thin, main, thin, big ? BRICK_BIG : BRICK_MAIN);
    const int64_t lo_offset = (int64_t)27 * (C1 + C2) * Cout;  // elements between the hi and the lo weight image
    const bool perm = main.v[0].perm[0] != 0;  // forward on ragged grids: the short brick edge on another axis
    auto go_main = [&]() -> int {
        if (NT == 2) {
            if (zero_pad && perm) { return conv3_mfma_split_go_2zmp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            if (zero_pad) { return conv3_mfma_split_go_2zmn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            if (perm) { return conv3_mfma_split_go_2rmp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            return conv3_mfma_split_go_2rmn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st);
        }
        if (big) {
            if (zero_pad && perm) { return conv3_mfma_split_go_1zbp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            if (zero_pad) { return conv3_mfma_split_go_1zbn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            if (perm) { return conv3_mfma_split_go_1rbp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
            return conv3_mfma_split_go_1rbn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st);
        }
        if (zero_pad && perm) { return conv3_mfma_split_go_1zmp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
        if (zero_pad) { return conv3_mfma_split_go_1zmn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
        if (perm) { return conv3_mfma_split_go_1rmp(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
        return conv3_mfma_split_go_1rmn(x1, C1, x2, C2, wp, bias, y, main, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st);
    };
    int rc = go_main();
    if (rc != TDX_OK) return rc;
    // remainder slabs of the padded grid: 2 x 16 x 8 bricks, 32-wide tiles
    if (thin.n > 0) { return conv3_mfma_split_go_1ztp(x1, C1, x2, C2, wp, bias, y, thin, Cout, lo_offset, gn_acc, d1, D1, d2, a1, a2, st); }
    return TDX_OK;
}
