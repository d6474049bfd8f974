// Buffers and hyper-parameters shared by the per-cell kernels (cells.hip, stn.hip, render.hip,
// loss.hip) and the host orchestration (engine.hip).
#pragma once
#include "layout.h"

// per-row statistics kept for the backward pass / KL
#define SP_LDSTAT 16
#define ST_MU_BOX 0    // 4: cy, cx, height, width (post-freeze)
#define ST_SD_BOX 4    // 4
#define ST_MU_DEPTH 8
#define ST_SD_DEPTH 9
#define ST_PZ 10       // count-prior probability p(z_pres=1 | counts so far)

struct CellHyper {
    float wheel;              // LATENT_VAR_TRAINING_WHEEL (0 or 1 with the reference config)
    float kl_scale;           // 1 / (B * world_size)
    float img;                // image side I (float)
    float anchor;             // ANCHORBOX_SHAPE[0]
    float cell_over_img;      // pixels_per_cell / I  (computed in double, rounded once: models.py:373)
    float max_yx, min_yx, max_hw, min_hw;
    float prior_mean[6], prior_std[6];   // cy, cx, height, width, attr, depth
    float count_prior_prob;   // sigmoid(log(v+1e-6)), models.py:186-188
    float range_yx, range_hw; // max - min of the two box ranges (fp32 differences, formed on the host: the chain kernels take them as scalar
                              // operands -- formed in the kernel the compiler hoists them into vector registers the forward chain does not have)
};

struct CellBufs {
    // tables (device)
    const int *cell_h, *cell_w, *cidx, *nbr, *cons, *diag_start;
    // inputs
    const float* feat; int ld_feat;
    const float* edge;
    const float *eps_box, *eps_attr, *eps_depth, *u_pres;   // NCHW maps [B,{4,A,1,1},G,G]
    const float* gloss;                                     // device scalar dL/dloss
    // forward activations (row-major, row r = cprime*B + b)
    float *Xb, *Hb1, *Hb2, *Ob;
    float *glimpse, *He1, *He2, *Oe;
    float *Xz, *Hz1, *Hz2, *Oz;
    float *Xo, *Ho1, *Ho2, *Oo;
    float *rec, *sd_attr, *nbox, *stat;
    float* Za;                                              // attr rows for the decoder [N, ld_rec], zero padded
    void *Za16, *dfeat16;                                   // fused chain (bf16 step): the same two tensors written directly as bf16
    unsigned long long* mbits;                              // fused chain only: relu sign bits of the 8 hidden layers, [B][3G-2][66 tiles][4] wave ballots
    unsigned int* gxy;                                      // fused chain only: per glimpse element (d val/d gx, d val/d gy) as fp16x2 [N, ld_gl]
    float *z_where, *z_pres;                                // NCHW outputs
    // backward
    float *dXb, *dHb1, *dHb2, *dOb;
    float *dGl, *dHe1, *dHe2, *dOe;
    float *dXz, *dHz1, *dHz2, *dOz;
    float *dXo, *dHo1, *dHo2, *dOo;
    float *grec, *g_nbox_stn;
    float *g_nbox_r, *g_pres_r, *g_depth_r, *g_attr_r;      // from the renderer / decoder
    float* dfeat;
};

int cells_init_tables(int G, int LB, int* cell_h, int* cell_w, int* cidx, int* nbr, int* cons, int* diag_start, hipStream_t s);
int cells_ctx_gather(const CellLayout& L, const CellBufs& P, int r0, int R, hipStream_t s);
int cells_box_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_attr_sample(const CellLayout& L, const CellBufs& P, int r0, int R, hipStream_t s);
int cells_depth_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_pres_sample(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_bwd_pres(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_bwd_depth(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_bwd_attr(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_bwd_box(const CellLayout& L, const CellBufs& P, const CellHyper& H, int r0, int R, hipStream_t s);
int cells_dfeat_edge(const CellLayout& L, const CellBufs& P, float* gedge, hipStream_t s);
