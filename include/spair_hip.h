/* libspair_hip.so -- C ABI of the MI355X-native SPAIR training step.
 *
 * This is the drop-in boundary beneath the Python surface `spair_pytorch_amd.models.SPAIR`
 * (which mirrors /root/reference/spair/models.py:15-131).  The reference has no FFI layer of
 * its own: what these entry points replace is the ATen/cuDNN/cuBLAS work its forward/backward
 * dispatches (SURVEY.md §2 "Library-op inventory").  Conventions:
 *   - every pointer is a DEVICE pointer into caller-owned memory; nothing is allocated or freed,
 *     scratch is passed in explicitly (`workspace`, zero-initialised ONCE by the caller: the
 *     library relies on never-written pad columns staying zero);
 *   - work is enqueued on `stream` (a hipStream_t) and never synchronises;
 *   - return 0 on success, a negative SPAIR_ERR_* code otherwise (shape / dtype / launch);
 *   - all step state lives in the buffers the caller passes.  The only library-owned objects are one low-priority helper HIP stream
 *     (+ 6 fork/join events) per device, created under a lock on first use or by spair_init(), and the opt-in profiling event pool
 *     (spair_prof_*, lock-protected, off by default).  Calls on different devices or different caller streams may be issued concurrently
 *     from different host threads as long as they use different workspaces: the fork/join events are per device, so each
 *     spair_forward / spair_backward holds that device's enqueue lock from its first fork to its last join (host-side only -- the
 *     enqueued work of the two callers still overlaps on the GPU; tests/test_surface_gpu.py drives two models from two threads).
 *     Create the helper stream with spair_init() before capturing a step into a hipGraph.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPAIR_DTYPE_F32 0   /* GEMM/conv operands fp32 (v_mfma_f32_16x16x4_f32), exact */
#define SPAIR_DTYPE_BF16 1  /* GEMM/conv operands bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16) */

/* Hyper-parameters = /root/reference/spair/config.py:3-76 plus the batch geometry. */
typedef struct SpairDims {
    int B, C, I, G;            /* batch, image channels (config.py:4 INPUT_IMAGE_SHAPE[0]: 1 = the benchmarked fused kernels; 2, 3 = colour
                                * images on the per-wavefront launches with the generic-channel renderer, either dtype), image side, grid side */
    int P, A, F, NP;           /* OBJECT_SHAPE[0], N_ATTRIBUTES, N_BACKBONE_FEATURES, N_PASSTHROUGH_FEATURES */
    int n_conv;                /* backbone conv layers before conv_out (config.py:7-14) */
    int conv_k[8], conv_s[8], conv_c[8];
    int pad_pre, pad_post, cell_px;   /* receptive-field padding (modules.py:68-105) */
    int dtype;                 /* SPAIR_DTYPE_* */
    int align_corners;         /* 0 = torch>=1.3 default (the pinned oracle), 1 = torch 1.0 era */
    float anchor;              /* ANCHORBOX_SHAPE[0] */
    float max_yx, min_yx, max_hw, min_hw;
    float obj_logit_scale, alpha_logit_scale, alpha_logit_bias;
    float vae_beta;            /* VAE_BETA (config.py:55) */
    float prior_mean[6], prior_std[6];   /* cy, cx, height, width, attr, depth (config.py:45-52) */
    /* Convolutional object encoder / decoder variant (CONV_OBJECT_ENCODER_TOPOLOGY, config.py:15-20; models.py:606-665 sketches the two
     * classes but cannot run them: PARITY UNPINNED).  obj_conv = 1 replaces the MLP encoder by oc_n valid convolutions (filters oc_c,
     * kernel oc_k, stride oc_s, ReLU after each) + Linear(flattened (C,H,W) -> 2A), and the MLP decoder by Linear(A -> flattened) + the
     * mirrored ConvTranspose2d stack (output_padding retraces the encoder's sizes; ReLU between, none after the last; its C+1 output
     * channels are the sprite's (colour.., alpha) logits).  Per-wavefront launches in either dtype; the convolutions themselves and the
     * sprites they produce are fp32 (the bf16 step keeps bf16 GEMM operands for the backbone, the other per-cell nets and the two Linears).  Parameter
     * names: object_encoder.conv.conv_<i>.{weight,bias}, object_encoder.out.*, object_decoder.inp.*,
     * object_decoder.conv.conv_transposed_<i>.* (ConvTranspose2d layout [in][out][k][k]). */
    int obj_conv, oc_n;
    int oc_k[4], oc_s[4], oc_c[4];
    /* N_LOOKBACK (config.py:31, models.py:292-320): a cell's lateral context is the 2L(L+1) already-visited cells of rows h-L..h,
     * columns w-L..w+L, in the reference's order (row-major, the current cell and those to its right dropped); out-of-grid slots read
     * the learned edge element.  0 is read as 1 (the reference's configuration: UL, U, UR, L).  L != 1 runs on the per-wavefront
     * launches (dependency wavefronts t = (L+1) h + w); the fused per-cell kernels are built for L = 1.  1 <= L <= 3. */
    int lookback;
} SpairDims;

/* Per-step scalars (host side evaluates the two schedules, modules.py:191-213). */
typedef struct SpairStep {
    float wheel;               /* LATENT_VAR_TRAINING_WHEEL value */
    float count_prior_prob;    /* 1/(1+exp(-log(v+1e-6))), models.py:186-188 */
    float kl_scale;            /* 1/(B*world_size): batch-mean of the KL terms (models.py:553) */
    int train;                 /* 1: keep what backward needs */
    int flags;                 /* bit 0: disable the fused persistent per-cell kernels (A/B testing); bit 1: record stage stamps;
                                * bit 2: no helper stream (every kernel on the caller's stream);
                                * bit 3: stem weight gradient as its own kernel (not fused into conv_1's data gradient);
                                * bit 4: decoder forward as three GEMM launches instead of the fused activation-stationary kernel;
                                * bit 5: strided backbone convs through the implicit-GEMM kernel instead of the patch-resident one;
                                * bit 6: decoder data gradients as three GEMM launches instead of the fused kernel */
    int draw_noise;            /* spair_forward only: 1 = fill eps_box/eps_attr/eps_depth/u_pres from noise_seed first (what spair_noise_fill
                                * does, but on the helper stream beside the backbone); the buffers must be writable */
    unsigned long long noise_seed;
    /* Non-finite / failed steps made loud without a host synchronisation (the reference RAISES on any NaN in its forward:
     * spair/debug_tools.py:245-271, called at models.py:65,108,245).  spair_forward's loss kernel evaluates
     *   bits = 1 * (a band-split hand-off of the per-cell chain ever timed out on this workspace) | 2 * (a loss term of THIS forward is NaN / inf)
     * and, where the pointers are non-null, writes
     *   status[0] |= bits (sticky), status[1] = bits (this step; spair_adam_guarded's skip word)   -- two ints in device memory, caller-owned;
     *   *status_host = bits, only when bits != 0   -- one int of host memory the device can write (spair_host_word_alloc), so that the
     *   caller can poll it with a plain load at any later point (a normal step never touches it).  Both may be NULL. */
    int* status;
    int* status_host;
} SpairStep;

/* Optional: create the current device's helper stream now (otherwise on the first spair_forward / spair_backward). */
int spair_init(void);

/* ---- parameter / workspace layout -------------------------------------------------------- */
/* Flat fp32 parameter buffer; tensor i of the reference state_dict (same key, same shape). */
int spair_param_count(const SpairDims* d);
int spair_param_info(const SpairDims* d, int idx, char* name, int name_cap, int64_t* offset, int64_t* shape4, int* ndim);
int64_t spair_param_total(const SpairDims* d);
int64_t spair_workspace_bytes(const SpairDims* d);

/* ---- the training step ---------------------------------------------------------------------
 * forward  == SPAIR.forward (models.py:35-131): backbone -> per-cell loop -> KL -> render -> loss.
 *   loss_out (>= 10 floats): [0]=total, [1]=BCE sum, [2..8]=KL cy,cx,height,width,attr,depth,pres (batch means), [9]=total again (a
 *   second copy for a host layer that hands the loss out as a view: an in-place op on it then leaves the logged terms [0..8] alone).
 * backward == loss.backward() (train.py:66): accumulates into `grads` (same layout as params).
 * Noise maps are NCHW: eps_box[B,4,G,G] (cy,cx,height,width), eps_attr[B,A,G,G],
 * eps_depth[B,1,G,G], u_pres[B,1,G,G]. */
int spair_forward(const SpairDims* d, const SpairStep* st, const float* params, const float* x,
                  const float* eps_box, const float* eps_attr, const float* eps_depth, const float* u_pres,
                  void* workspace, float* loss_out, float* recon, float* z_where, float* z_pres, void* stream);
int spair_backward(const SpairDims* d, const SpairStep* st, const float* params, const float* x,
                   const float* eps_box, const float* eps_attr, const float* eps_depth, const float* u_pres,
                   void* workspace, const float* grad_loss, float* grads, void* stream);
/* Data-parallel hook (the reference is single-device, train.py:27-30): the backward completes the gradients in three
 * contiguous ranges of `grads` -- [0] decoder, [1] box/encoder/z/obj nets, [2] edge element + backbone, in that order.
 * spair_grad_buckets returns the ranges (element offsets); spair_backward_ev is spair_backward that additionally records the
 * caller-created hipEvent_t ev_* (null = skip) when the corresponding range is final, so its all-reduce can overlap the rest. */
int spair_grad_buckets(const SpairDims* d, int64_t* lo3, int64_t* hi3);
int spair_backward_ev(const SpairDims* d, const SpairStep* st, const float* params, const float* x,
                      const float* eps_box, const float* eps_attr, const float* eps_depth, const float* u_pres,
                      void* workspace, const float* grad_loss, float* grads, void* stream,
                      void* ev_decoder, void* ev_cells, void* ev_backbone);
/* torch.optim.Adam(lr) defaults (train.py:44) on flat buffers, one launch. */
int spair_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
               float beta1, float beta2, float eps, int step, void* stream);
/* The same update, guarded: if `skip` (device int, e.g. SpairStep.status + 1) is non-zero the whole step is left out -- parameters and
 * both moments untouched -- and counters[0] is incremented; an element whose gradient is NaN / inf is left out on its own and counters[1]
 * is set to 1.  lr * NaN never reaches a parameter.  `counters`: two device ints, caller-owned and caller-zeroed; the caller decrements its
 * bias-correction step for a skipped step if it wants torch's numbers after a recovery (FusedAdam reads the counter where it
 * synchronises anyway).  skip may be NULL (element guard only). */
int spair_adam_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, int step, const int* skip, int* counters, void* stream);
/* One int of host memory that kernels can store to (hipHostMalloc, mapped + coherent), zero-initialised: SpairStep.status_host. */
int spair_host_word_alloc(int** out);
int spair_host_word_free(int* word);
/* Copy a per-row quantity of the last forward into an NCHW map [B,ch,G,G].
 * which: 0 z_attr, 1 z_depth, 2..7 mean of cy,cx,height,width,attr,depth, 8..13 their sigma, 14 count-prior p_z;
 * after a backward, its per-cell latent gradients: 100 d box head latents [8] (mean 4 | log-std 4), 101 d encoder output [2A], 102 d depth
 * latents [2], 103 d presence logit [1] as the per-wavefront launches store them (fp32 rows); 200..203 the same as the fused chain stores
 * them (bf16 rows) */
int spair_export_map(const SpairDims* d, const void* workspace, int which, float* out, void* stream);
/* diagnostic: stage time stamps of the fused forward chain kernel (SpairStep.flags bit 1), n <= 4096 uint64 */
int spair_chain_stamps(const SpairDims* d, const void* workspace, unsigned long long* out, int n, void* stream);
/* layout of that buffer: stamps per wavefront of the forward kernel (from offset 0; stage intervals = stamps - 1), the index of the glimpse
 * sampling interval (K4: modules.py:216-273 via models.py:387) among them, stamps per wavefront of the backward kernel (from offset 2048) */
int spair_chain_stamp_layout(int* fwd_per_wavefront, int* fwd_glimpse_interval, int* bwd_per_wavefront);
/* wavefronts walked by the workgroup that stamps (sample 0; with the band split of grids wider than 16 cells, its top band) */
int spair_chain_stamp_wavefronts(const SpairDims* d);
/* band split of the fused per-cell kernels (grids wider than 16 cells: ceil(G / 8) workgroups per sample hand the boundary rows' records /
 * context gradients to each other through `workspace`): writes 1 to *out (device int) if a bounded wait EVER timed out in a
 * spair_forward / spair_backward on this workspace (sticky: only re-zeroing the workspace clears it; from that step on loss_out[0] and the
 * gradient of virtual_edge_element are NaN, so a training loop sees it without calling this), 0 if not, -1 where the kernels run unsplit */
int spair_chain_sync_status(const SpairDims* d, const void* workspace, int* out, void* stream);
int spair_noise_fill(const SpairDims* d, uint64_t seed, float* eps_box, float* eps_attr, float* eps_depth, float* u_pres, void* stream);

/* Opt-in instrumentation for bench.py: HIP events on the caller's stream around regions of the step.
 * slots: 0 prep, 1 backbone fwd, 2 per-cell chain fwd, 3 decoder fwd, 4 count-prior KL, 5 render fwd (1 kernel),
 * 6 KL+loss, 7 render bwd (1 kernel), 8 decoder bwd, 9 per-cell chain bwd, 10 per-cell weight grads,
 * 11 backbone bwd, 12 conv_1 fwd (1 kernel), 13 decoder.out fwd GEMM (1 kernel), 14 STN glimpse fwd (per wavefront),
 * 15 adam, 16 decoder.out wgrad, 17 decoder.out dgrad.  spair_prof_read synchronises: call it outside timed regions. */
int spair_prof_enable(int enable);                      /* 0 stop, 1 start afresh, 2 resume (keeps earlier records) */
int spair_prof_select(unsigned long long slot_mask);   /* record only the regions whose bit is set (default: all) */
int spair_prof_read(float* ms, int* counts, int nslots);

/* ---- unit-level entry points (each kernel can be parity-checked alone) --------------------- */
int spair_gemm_nt(const float* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                  const float* bias, const float* relu_mask, int ldmask, int relu, int accumulate, int dtype,
                  void* stream);
int spair_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R,
                  int dtype, void* stream);
int spair_gemm_nt_conv(const float* In, const int* conv13, const void* B, int ldb, float* C, int ldc, int M,
                       int N, int K, const float* bias, const float* relu_mask, int ldmask, int relu,
                       int accumulate, const int* cmap8, int dtype, void* stream);
int spair_gemm_tn_conv(const float* A, int lda, const float* In, const int* conv13, float* C, int ldc, int M,
                       int N, int R, int dtype, void* stream);
int spair_colsum(const float* A, int lda, int R, int N, float* out, void* stream);
/* Direct fp32 convolutions of the convolutional object encoder / decoder variant (objconv.hip; SpairDims.obj_conv).  A tensor is
 * described by t6 = {rs, ys, xs, cs, H, C}: element (r, y, x, c) at p[r*rs + y*ys + x*xs + c*cs], H x H pixels, C channels.  W is
 * [X][Y][k][k] (nn.Conv2d: X = out, Y = in channels; nn.ConvTranspose2d: X = in, Y = out channels).
 * spair_objconv_gather, transposed = 0: out(r,y,x,X) = bias + sum in(r, y*s+ky, x*s+kx, Y) * W  (Conv2d forward / ConvTranspose2d data
 *   gradient); transposed = 1: out(r,y,x,Y) = bias + sum in(r, (y-ky)/s, (x-kx)/s, X) * W  (ConvTranspose2d forward / Conv2d data
 *   gradient); then out = 0 where gate <= 0 (gate: same layout as out, or NULL), then ReLU if relu.
 * spair_objconv_wgrad: G[cs][cb][ky][kx] += sum small(r,y,x,cs) * big(r, y*s+ky, x*s+kx, cb), bias_small[cs] += sum small (or NULL)
 *   (Conv2d: small = d out, big = in; ConvTranspose2d: small = in, big = d out). */
int spair_objconv_gather(int transposed, const float* in, const long long* in6, const float* W, const float* bias, float* out,
                         const long long* out6, const float* gate, int k, int s, int relu, long long R, void* stream);
int spair_objconv_wgrad(const float* small, const long long* small6, const float* big, const long long* big6, float* G,
                        float* bias_small, int k, int s, long long R, void* stream);
/* bf16-STORED operand GEMMs (the bf16 mode's activations and gradients live in HBM as bf16; same roles as above).
 * spair_gemm_nt16: C = epi(A * B^T), A bf16 [M][lda] or an NHWC conv gather (conv13), B bf16 [N][ldb], C bf16 (c_bf16)
 *   or fp32; relu_mask bf16 (mask_bf16) or fp32; cmap8 remaps output rows (stride-2 conv data gradient by parity class).
 * spair_gemm_tn16: C += A^T * B over R rows, A bf16 [R][lda], B bf16 rows / bf16 or fp32 conv gather; cw_cin/cw_taps store
 *   columns in OIHW order; colsum_out += column sums of A (bias gradient).  scratch (optional, >= blocks*128*128 floats):
 *   split-K partial tiles + a reduce pass instead of fp32 atomics (modules.py:59-64,124-165 autograd). */
int spair_gemm_nt16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                    const float* bias, const void* relu_mask, int ldmask, int mask_bf16, int relu, int c_bf16,
                    const int* conv13, const int* cmap8, void* stream);
int spair_gemm_tn16(const void* A, int lda, const void* B, int ldb, int b_bf16, float* C, int ldc, int M, int N,
                    int R, const int* conv13, int cw_cin, int cw_taps, float* colsum_out, float* scratch,
                    long long scratch_floats, void* stream);
/* weight + bias gradient of the single-channel 4x4 stem conv with 128 filters (Backbone layer 0, modules.py:59-64):
 * dY bf16 [B*Hout*Hout][128], xpad fp32 [B][Hin][Hin]; dW [128][1][4][4] and db [128] are accumulated;
 * scratch >= 512*128*32 floats */
int spair_stem_wgrad16(const void* dY, const float* xpad, float* dW, float* db, float* scratch,
                       long long scratch_floats, int B, int Hin, int stride, int Hout, void* stream);
/* Fused stack of L <= 4 1x1 convolutions on bf16 NHWC activations with 128 channels (Backbone's trailing 1x1 layers + conv_out,
 * modules.py:59-64,107-111).  Host arrays of L device pointers.
 * fwd: Y_l = relu(Y_{l-1} W_l^T + b_l); W[l] bf16 [cout_l][ldw_l]; Y[l] bf16 [M][128] for l < L-1; the last layer has no relu and
 *      writes fp32 Ylast [M][ldlast] (cout_{L-1} <= 128 columns).
 * bwd: layers in BACKWARD order; dX_l = (dX_{l-1} Wd_l^T) * [gate_l > 0]; dY bf16 [M][ldd] (kd valid columns), Wd[l] bf16
 *      [128][ldw_l] = W transposed, gate[l] = the layer's forward INPUT (bf16 [M][128]), dX[l] bf16 [M][128]. */
int spair_conv1x1_stack_fwd16(const void* X, const void* const* W, const int* ldw, const int* cout,
                              const float* const* bias, void* const* Y, float* Ylast, int ldlast, int M, int L,
                              void* stream);
int spair_conv1x1_stack_bwd16(const void* dY, int ldd, int kd, const void* const* Wd, const int* ldw, const int* cout,
                              const void* const* gate, void* const* dX, int M, int L, void* stream);
/* ---- evaluation metrics (spair/metric.py:5-99), device-side, no in-place mutation ------------------------------ */
/* z_where [B,4,G,G] (x, y, w, h image fractions, the reference's top-left convention), z_pres [B,1,G,G], bbox [B,K,4]
 * (x, y, w, h px, zero padded), count [B] fp32; scratch 2*B floats; out[0] = mAP (metric.py:5-47),
 * out[1] = object_count_accuracy (metric.py:49-56) */
int spair_metrics(const float* z_where, const float* z_pres, const float* bbox, const float* count, int B, int G,
                  int image_side, int K, float* scratch, float* out, void* stream);
/* batch_jaccard (metric.py:82-99): corner-format boxes [B,A,4] x [B,Bn,4] -> iou [B,A,Bn] */
int spair_batch_jaccard(const float* box_a, const float* box_b, int B, int A, int Bn, float* iou, void* stream);
/* ---- synthetic scattered-digit scenes generated on the device (stands in for spair/dataloader.py:10-36, whose HDF5 file is not
 * available; same item contract).  image [B,1,I,I] fp32 in [0,1], bbox [B,K,4] fp32 (x, y, w, h px, zero padded), count [B] int64;
 * samples first..first+B-1 of the Philox stream `seed`; scratch B*K*28 floats. */
int spair_scenes_generate(uint64_t seed, long long first, int B, int I, int K, int size_min, int size_max, float* image,
                          float* bbox, long long* count, float* scratch, void* stream);
/* fp32 [rows][ld_src] -> bf16 [rows][ld_dst] (round to nearest even), first `cols` columns */
int spair_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, long long rows, int cols, void* stream);
/* stn(image, z_where, [P,P]) forward (border) and its gradient wrt z_where (modules.py:216-273);
 * row r samples image x[r % B], nbox[r] = (xt,yt,xs,ys) */
int spair_stn_glimpse_fwd(const float* x, const float* nbox, int B, float* glimpse, int ld_gl, int R, int C,
                          int I, int P, int align_corners, void* stream);
int spair_stn_glimpse_bwd(const float* x, const float* nbox, int B, const float* dglimpse, int ld_gl,
                          float* dnbox, int R, int C, int I, int P, int align_corners, void* stream);
/* stn(sprites, z_where, [I,I], inverse=True) materialised (modules.py:256-269): sprites [N,C,P,P] -> out [N,C,I,I], bilinear, zeros
 * padding, inverse affine in closed form; backward ACCUMULATES into dsprites [N,C,P,P] and dnbox [N,4] (zero them first).  Only for
 * callers of the reference's helper -- the training step never materialises this tensor (spair_render_fwd fuses it). */
int spair_stn_inverse_fwd(const float* sprites, const float* nbox, float* out, int N, int C, int P, int I, int align_corners,
                          void* stream);
int spair_stn_inverse_bwd(const float* sprites, const float* nbox, const float* grad_out, float* dsprites, float* dnbox, int N,
                          int C, int P, int I, int align_corners, void* stream);
/* renderer: inverse STN + importance-weighted composite + BCE (models.py:485-547) */
int spair_render_fwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                     const float* x, float* recon, float* aux /* B*I*I float2: (dBCE/dpre / D, pre) */, float* bce_partial,
                     int B, int HW, int C, int I, int P, int align_corners, void* stream);
int spair_render_bwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                     const float* aux, const float* grad_loss, float* dlogits, float* dnbox, float* dpres, float* ddepth,
                     int B, int HW, int C, int I, int P, int align_corners, float obj_scale, float alpha_scale, void* stream);
/* The same for images with C = 2 or 3 colour channels (cfg.INPUT_IMAGE_SHAPE[0], models.py:480,524; render_c.hip): sprites fp32
 * [N][ld_s] = [P*P][C+1] (colour.., alpha) after the sigmoid, x / recon [B][C][I][I], aux B*C*I*I float2, dlogits fp32 [N][ld_s].
 * Generic-channel kernels (correctness and run-to-run determinism; the tuned renderers are single-channel). */
int spair_render_fwd_rgb(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth, const float* x,
                         float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int align_corners, void* stream);
int spair_render_bwd_rgb(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth, const float* aux,
                         const float* grad_loss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int B, int HW, int C, int I,
                         int P, int align_corners, float obj_scale, float alpha_scale, void* stream);
/* Backbone stem alone: conv 1 -> Cout channels, 4x4, stride `stride`, no padding, + bias + relu, over the image zero-padded to Hin x Hin
 * (pad_pre pixels before; modules.py:95-104 Backbone.padding + the first Conv2d/ReLU of Backbone.net).  x [B][I][I] fp32 (unpadded),
 * w [Cout][16], out NHWC [B][Hout][Hout][Cout] fp32 or (out_bf16) bf16.  In bf16 mode with Cout = 128, stride 2 it runs on the matrix
 * cores with split-bf16 operands (three products, fp32 accumulation): agrees with the fp32 result to < 2^-15 relative before the store. */
int spair_stem_conv_fwd(const float* x, const float* w, const float* bias, void* out, int B, int I, int pad_pre, int Hin, int Hout,
                        int Cout, int stride, int out_bf16, void* stream);
/* Patch-resident forward of the backbone's 128 -> 128 channel, 4x4, stride-2 convolutions + bias + ReLU (reference modules.py:59-64), bf16 NHWC in / out
 * (csrc/conv_s2.hip).  in16 [B][Hin][Hin][128] with Hin = 2 * Hout + 2 (the input is pre-padded); wf16 [128][2048] bf16 in tap-parity K order:
 * column ((class * 2 + half) * 4 + tap) * 64 + c holds W[o][ci = half * 64 + c][ky = py + 2 dy][kx = px + 2 dx], class = 2 py + px, tap = 2 dy + dx;
 * out16 [B * Hout * Hout][128].  Returns SPAIR_ERR_UNSUPPORTED when a 256-row tile's input patch exceeds the kernel's LDS buffer. */
int spair_conv_s2k4_fwd16(const void* in16, const void* wf16, const float* bias, void* out16, int B, int Hin, int Hout, void* stream);
/* Patch-resident DATA GRADIENT of the same layers (csrc/conv_s2_dgrad.hip): dout16 bf16 NHWC [B][Ho][Ho][128]; wdq: bf16 [128 ci][4 * 128], column
 * (ty * 2 + tx) * 128 + co = W[co][ci][py + 2 ty][px + 2 tx] for output-parity class q = 2 py + px; gate16: the stored activation of the layer
 * below, bf16 NHWC [B][2 (Ho + 1)][2 (Ho + 1)][128]; out16 (same shape) = conv2d_backward_input(dout, W) where gate16 > 0, else 0. */
int spair_conv_s2k4_dgrad16(const void* dout16, const void* wd0, const void* wd1, const void* wd2, const void* wd3, const void* gate16,
                            void* out16, int B, int Ho, void* stream);
/* Sign-bit form of a ReLU gate (round 5): one byte per (pixel, 8 channels), [B][H][H][16], bit e = channel 8 g + e of the stored bf16
 * activation > 0.  spair_stem_conv_fwd_mask: the stem (1 -> 128 channels, 4 x 4, stride 2, bf16 output) leaving that mask beside its output;
 * spair_conv_s2k4_dgrad16_bits: spair_conv_s2k4_dgrad16 reading the gate from it (20 MB instead of the 321-MB activation at the benchmark
 * shape) -- what the training step runs for conv_1's data gradient. */
int spair_stem_conv_fwd_mask(const float* x, const float* w, const float* bias, void* out, void* mask8, int B, int I, int pad_pre, int Hin,
                             int Hout, void* stream);
int spair_conv_s2k4_dgrad16_bits(const void* dout16, const void* wd0, const void* wd1, const void* wd2, const void* wd3,
                                 const void* gate_bits8, void* out16, int B, int Ho, void* stream);
/* The bf16 step's object-decoder FORWARD (reference models.py:474-492: Linear 50->128, ReLU, Linear 128->256, ReLU, Linear 256->P*P*2, the sprite
 * scales and analytic sigmoid) as one activation-stationary kernel (csrc/dec_fused.hip).  z_attr16: bf16 [N][ld_za] (columns >= A ignored);
 * W*, b*: the fp32 parameters, row-major [out][in]; H1 / H2: bf16 [N][128] / [N][256] hidden activations (stored for the backward);
 * sprites: fp16 [N][ld_s] (grey, alpha) pairs; stream_buf: spair_decoder_fwd16_scratch_bytes(n_out) bytes of scratch (the packed weights). */
int64_t spair_decoder_fwd16_scratch_bytes(int n_out);
int spair_decoder_fwd16(const void* z_attr16, int ld_za, const float* W0, const float* b0, const float* W1, const float* b1,
                        const float* W2, const float* b2, void* H1, void* H2, void* sprites, int ld_s, long long N, int A, int n_out,
                        float obj_scale, float alpha_scale, float alpha_bias, void* stream_buf, void* stream);
/* The decoder's DATA-GRADIENT chain (autograd of the three Linear layers of models.py:474-484 w.r.t. their inputs; csrc/dec_fused_bwd.hip) in one
 * launch.  dlogits16: bf16 [N][ld_s] (n_out columns, the sigmoid's derivative already applied); W2t16 / W1t16 / W0t16: the TRANSPOSED weights as
 * bf16, [256][ld2], [128][256], [A][128]; H2 / H1: the stored forward activations bf16 [N][256] / [N][128] (relu gates); outputs: dH2 / dH1 (bf16,
 * same shapes) and d_z_attr fp32 [N][ld_dza] (A columns written).  A <= 64, n_out % 8 == 0. */
int spair_decoder_bwd16(const void* dlogits16, int ld_s, const void* W2t16, int ld2, const void* W1t16, const void* W0t16, const void* H2,
                        const void* H1, void* dH2, void* dH1, float* d_z_attr, int ld_dza, long long N, int A, int n_out, void* stream);
/* the same with 16-bit sprites, as the bf16 training step runs them: sprites are FP16 (grey, alpha) pairs [N][ld_s] (post-sigmoid values
 * in (0,1): 11 significant bits), d-logits come back as BF16 [N][ld_s] */
int spair_render_fwd16(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth,
                       const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P,
                       int align_corners, void* stream);
int spair_render_bwd16(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth,
                       const float* aux, const float* grad_loss, void* dlogits_bf16, float* dnbox, float* dpres, float* ddepth,
                       int B, int HW, int C, int I, int P, int align_corners, float obj_scale, float alpha_scale, void* stream);
/* The forward renderer of the bf16 step on the matrix cores (csrc/render3.hip; same reference lines: stn(inverse=True) modules.py:256-269 +
 * the composite models.py:511-540).  The inverse-STN sampling is separable, out_c = Wy . S_c . Wx^T with hat weights, and runs as three
 * v_mfma_f32_16x16x32_f16 per channel and (object, 16 x 16 tile) on the fp16 sprites as they lie in memory.
 * spair_render_prep writes 64 bytes of records per object (source-coordinate coefficients, presence, importance scale / floor, pixel footprint,
 * raw inverse-affine parameters; sample-major, 64 * B * HW bytes, 16-byte aligned, caller-owned) from the rows r = k * B + b of nbox [N][4] / pres [N] / depth [N];
 * spair_render_fwd16m composites from the records: same outputs as spair_render_fwd16 (recon, aux, bce_partial), per pixel within 5e-4 of
 * it (fp16 hat weights on source coordinates rounded to 2^-11 texel, one fp16 rounding of the x-interpolated rows), unbiased.
 * SPAIR_ERR_UNSUPPORTED for P != 28, align_corners, HW > 1024: use spair_render_fwd16. */
int spair_render_prep(const float* nbox, const float* pres, const float* depth, void* records, int B, int HW, int I, int P,
                      int align_corners, void* stream);
int spair_render_fwd16m(const void* sprites_f16, int ld_s, const void* records, const float* x, float* recon, float* aux,
                        float* bce_partial, int B, int HW, int C, int I, int P, int align_corners, void* stream);
/* spair_render_bwd16 with the inverse-affine parameters and pixel footprints read from the records of spair_render_prep (the same nbox /
 * pres / depth) instead of recomputed per object: what the training step runs; identical outputs */
int spair_render_bwd16r(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth, const void* records,
                        const float* aux, const float* grad_loss, void* dlogits_bf16, float* dnbox, float* dpres, float* ddepth,
                        int B, int HW, int C, int I, int P, int align_corners, float obj_scale, float alpha_scale, void* stream);
#ifdef __cplusplus
}
#endif
