#pragma once
#include "common.h"

struct GemmNT {
    const float* A; int lda;
    const void* B; int ldb;
    float* C; int ldc;
    int M, N, K;
    const float* bias;
    const float* mask; int ldmask;
    int relu, accumulate;
    int c_bf16, mask_bf16;                // gemm16.hip: C stored as bf16; relu mask source is a bf16 tensor
    int sprite_ch;                        // >0: decoder epilogue (models.py:485-492): analytic sigmoid of scaled logits, alpha = last of sprite_ch
    float obj_scale, alpha_scale, alpha_bias;
    ConvDesc conv;
    RowMap cmap; int use_cmap;
    // gemm16.hip: nz > 1 batches the nz = osy*osx output-parity classes of a strided conv data-gradient in ONE launch
    // (blockIdx.z = py*osx + px picks the weight matrix Bz[z] and the row-map offsets); all classes must have the same size.
    int nz; const void* Bz[4];
    // gemm16.hip conv gathers, optional: K runs over 64-column blocks in tap-parity order -- ktab[kt] = ky << 24 | kx << 16 | first channel
    // of K block kt (n_ktab = K / 64 <= 64 entries, Cin % 64 == 0, B stored in the same order).  The taps of one stride-parity class read
    // the same input lattice shifted by one output pixel, so running them back to back turns the 4x re-use of a strided conv's input
    // into L2 hits (conv_1 forward fetched 860 MB for a 321 MB input in (ky, kx, ci) order).
    int n_ktab; unsigned ktab[64];
    int xcd_tiles_n;                      // filled by the gemm16.hip launcher: > 1 = linear grid, the N tiles of an M block adjacent on one XCD
    // gemm16.hip, conv_1's data gradient only: stem_part != NULL fuses the stem's weight gradient into the epilogue (the gated
    // tile x the 4x4 patches of the padded single-channel input stem_xp [B][stem_hin][stem_hin], stride stem_s) and skips the C
    // store; per-workgroup partials go to stem_part (capacity in floats), their sum is added to stem_dw [128][16] / stem_db [128].
    const float* stem_xp; float* stem_part; long long stem_part_cap; float* stem_dw; float* stem_db; int stem_hin, stem_s;
};
bool spair_nt16_stem_fusable(const GemmNT& g, long long part_cap);
#define SPAIR_TN_MAX_TILES 40
struct GemmTN {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int M, N, R;            // M, N: load extents (multiples of 4, may run into zero pad columns)
    int Mstore, Nstore;     // store extents (<= M, N)
    int cw_cin, cw_taps;    // if cw_cin > 0: n = tap*cin + ci is stored at column ci*taps + tap (OIHW conv weights)
    int rows_per_split, nsplit, tiles_m, tiles_n;   // filled by the launcher
    float* colsum_out;      // optional: += column sums of A (bias gradient), fused into the same pass
    ConvDesc conv;
    // optional split-K scratch: each block stores its fp32 tile with plain coalesced stores to part[split][tile][128][128]
    // and a second kernel sums the splits and does C += sum (no atomics).  NULL / too small -> fp32 atomics into C.
    float* part; long long part_cap;    // capacity in floats
    float* colpart;                     // filled by the gemm16.hip launcher: per-block column sums [split][tile][128] behind the partial tiles
    // gemm16.hip grouped launch: ngroup independent single-tile problems (M, N <= 128 each) over the same R rows in ONE launch +
    // ONE reduce pass -- per-layer launches of small weight gradients are pure overhead (the per-cell nets' 14 GEMMs = 36 tiles took
    // 0.74 ms, the backbone's four 1x1 layers 4 x 70 us).  Operand pointers are pre-offset to the tile's first column.  Requires `part`.
    int ngroup;
    struct Tile {
        const void* A; const void* B; float* C; float* colsum;     // A bf16 [R][lda]; B bf16 or fp32 [R][ldb]; colsum may be null
        int lda, ldb, ldc;
        int M, N;                 // load extents (A: multiple of 8 columns; B: multiple of 8 (bf16) / 4 (fp32)), <= 128
        int Mstore, Nstore;       // store extents
        int m_skip, n_skip;       // the first m_skip rows / n_skip columns of the tile belong to another tile (operand pointers must stay
                                  // 16-byte aligned): tile row m is stored at C row m - m_skip for m_skip <= m < m_skip + Mstore, same for columns
    } tile[SPAIR_TN_MAX_TILES];
};
#define SPAIR_TN_PART_FLOATS (1536ll * 128 * 128)    // up to 1536 blocks x one 128x128 tile (100 MB)
int spair_gemm_nt_impl(const GemmNT& g, bool conv, int dtype, hipStream_t s);
int spair_gemm_tn_impl(GemmTN g, bool conv, int dtype, hipStream_t s);
int spair_tn_reduce(const GemmTN& g, int bm, int bn, hipStream_t s);   // second stage of a split-K TN GEMM that wrote g.part
int spair_colsum_impl(const float* A, int lda, int R, int N, float* out, hipStream_t s);
// gemm16.hip: bf16-stored operands
int spair_gemm_nt16_impl(const GemmNT& g, bool conv, hipStream_t s);
int spair_gemm_tn16_impl(GemmTN g, bool conv, bool b_bf16, hipStream_t s);
// tn_ring.hip: the same product through the DMA-staged split-K kernel (bf16 B, scratch required); SPAIR_ERR_UNSUPPORTED -> caller's kernel
int spair_gemm_tn_ring(GemmTN g, bool conv, hipStream_t s);
// stem (Cin = 1, 4x4, Cout = 128) weight + bias gradient; dY bf16 [B*Hout*Hout][128], xp padded fp32 input [B][Hin][Hin]
int spair_stem_wgrad16_impl(const void* dY, const float* xp, float* dW, float* db, float* part, long long part_cap, int B, int Hin,
                            int stride, int Hout, hipStream_t s);
// pointwise.hip: fused stacks of 1x1 convolutions on bf16 [M][128] activations (<= 4 layers, 128 channels, last/first layer <= 128)
int spair_pw_stack_fwd16(const void* X, const void* const* W, const int* ldw, const int* cout, const float* const* bias, void* const* Y,
                         float* Ylast, int ldlast, int M, int L, hipStream_t s);
int spair_pw_stack_bwd16(const void* dY, int ldd, int kd, const void* const* Wd, const int* ldw, const int* cout, const void* const* gate,
                         void* const* dX, int M, int L, hipStream_t s);
int spair_to_bf16(const float* src, int lds_, void* dst, int ldd, long long rows, int cols, hipStream_t s);
// conv_s2.hip: patch-resident forward of a 128 -> 128 channel 4x4 / stride-2 convolution (+ bias + relu), bf16 NHWC in / out, weights in
// tap-parity K order; SPAIR_ERR_UNSUPPORTED when the geometry does not fit (the caller keeps the implicit-GEMM kernel)
int conv_s2k4_patch_fwd16(const void* in, const void* wf, const float* bias, void* out, int B, int Hin, int Hout, int cin, int cout, int k, int s_,
                          hipStream_t s, void* mask = nullptr);      // mask: optional sign bits of the output [B*Hout*Hout][16] bytes
bool conv_s2k4_patch_fwd16_fits(int B, int Hin, int Hout, int cin, int cout, int k, int s_);
// conv_s2_dgrad.hip: patch-resident data gradient of the same layers (all 4 output-parity classes per workgroup), ReLU gate of the layer below,
// optionally with the stem's weight gradient fused (stem_part != nullptr: nothing is stored to `out`); gate_bits != nullptr: the gate as
// sign bits, one byte per (pixel, 8 channels) -- what the stem kernel leaves beside act0 (misc.hip) -- instead of the activation itself
int conv_s2k4_patch_dgrad16(const void* dout, const void* const* wd, const void* gate, void* out, int B, int Ho, int hin, int cin, int cout, int k,
                            int s_, const float* stem_xp, int stem_hin, int stem_s, float* stem_part, long long stem_part_cap, float* stem_dw,
                            float* stem_db, hipStream_t s, const void* gate_bits = nullptr);
