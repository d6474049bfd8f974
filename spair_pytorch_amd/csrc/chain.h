#pragma once
#include "cells.h"

enum ChainW { CW_BOX0, CW_BOX1, CW_BOXH, CW_ENC0, CW_ENC1, CW_ENC2, CW_Z0, CW_Z1, CW_ZH, CW_OBJ0, CW_OBJ1, CW_OBJ2, CW_COUNT };

struct ChainArgs {
    CellLayout L;
    CellBufs P;
    CellHyper H;
    const uint4* w[CW_COUNT];     // fragment-packed bf16 weights (k_prep mode 4 / 5)
    const uint4* wlo[3];          // forward only: the LOW parts w - bf16(w) of the box network's three layers (split-bf16 products, chain.hip)
    const float* bias[CW_COUNT];
    const uint4* wt[CW_COUNT];    // data-gradient packs (k_prep mode 5): B[k=out][n=in]
    const float* w_obj2;          // obj_network.out.weight [1,100] fp32 (rank-1 data-gradient)
    float* gedge;                 // gradient of virtual_edge_element
    float* gedge_part;            // [B][4][REC] per-(sample, neighbour slot) partials of it (k_chain_bwd writes, chain_edge_reduce sums them in sample order)
    unsigned long long* stamps;   // diagnostic: s_memtime after every stage of sample 0 (null in production)
    const float* x;
    int I, Pp, ac;
    // Band split (grids wider than 16 cells): a sample's grid rows are cut into `nbands` horizontal bands, one workgroup each.  A band only
    // ever READS from the band above it (a cell's context is (h-1, w-1), (h-1, w), (h-1, w+1), (h, w-1): models.py:297-304), so the bands of a
    // sample run as a pipeline -- the upper band publishes its last row's records, the lower one consumes them one wavefront later; the
    // backward kernel runs the same pipeline upwards with the context gradients.  `sync` (ints, zeroed before every launch): [0] start
    // ticket, [2] time-out flag, [16 + b*nbands + band] wavefronts published; behind them ONE word that no launch clears
    // (CHAIN_SYNC_STICKY: set with [2], zeroed only with the workspace) -- the loss and the edge gradient of a step turn NaN while it is set;
    // bnd_rec [B][nbands][G][REC], bnd_grad [B][nbands][G][3][REC].
    int nbands;
    int* sync;
    float* bnd_rec;
    float* bnd_grad;
};
#define CHAIN_MAX_BANDS 4
#define CHAIN_SYNC_HDR 16
#define CHAIN_SYNC_STICKY(B, nbands) (CHAIN_SYNC_HDR + 2 * (B) * (nbands))      // index of the sticky time-out word
#define CHAIN_SYNC_WORDS(B, nbands) (CHAIN_SYNC_STICKY(B, nbands) + 4)
int chain_bands(const SpairDims& d);          // 1 up to 16 x 16 cells; bands of 8 grid rows beyond (a wavefront then has <= 8 cells per band)

// stage stamps (diagnostic, SpairStep.flags bit 1): stamps per wavefront; the forward kernel's intervals are
// rows | S0 ctx | BOX0 | BOX1 | BOXH+box | glimpse | ENC0 | ENC1 | ENC2 | attr | Z0 | Z1 | ZH+depth | OBJ0 | OBJ1+obj2 | pres
#define CHAIN_FWD_STAMPS 17
#define CHAIN_FWD_GLIMPSE 5
#define CHAIN_BWD_STAMPS 19

int chain_fwd_supported(const SpairDims& d);
// 1 when the fused forward kernel samples glimpses from an fp16 LDS copy of the image (the per-wavefront reference path then
// rounds pixels the same way, so the two paths stay comparable to rounding level)
int chain_image_fp16(const SpairDims& d);
int chain_fwd(const ChainArgs& a, hipStream_t s);
int chain_bwd(const ChainArgs& a, hipStream_t s);
int chain_edge_reduce(const ChainArgs& a, hipStream_t s);     // after chain_bwd, any stream ordered behind it: gedge += sum_b gedge_part[b]
