// Host orchestration of the SPAIR training step behind the C ABI (include/spair_hip.h):
// workspace carving, per-step weight preparation, the backbone, the dependency-wavefront loop of
// the per-cell encoder (3G-2 dependent steps instead of the reference's G^2),
// decoder + renderer + KL, and the hand-written backward of all of it.
// Nothing here synchronises or allocates: every launch goes to the caller's stream.
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <mutex>
#include <functional>
#include <vector>

#include "cells.h"
#include "gemm.h"
#include "misc.h"
#include "chain.h"
#include "dec_fused.h"
#include "objconv.h"

// kernels in other translation units
int stn_glimpse_fwd(const float* x, const float* nbox, int B, float* out, int ld, int r0, int R, int C, int I, int P, int ac, int px16, hipStream_t s);
int stn_glimpse_bwd(const float* x, const float* nbox, int B, const float* dgl, int ld, float* dnbox, int r0, int R, int C, int I, int P, int ac, int px16, hipStream_t s);
int render_sprite_act(float* S, int ld, int N, int per, int CH, float obj_scale, float alpha_scale, float alpha_bias, hipStream_t s);
int render_num_blocks(int B, int I);
int render_fwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int ac, int s_bf16, hipStream_t s);
int render_prep(const float* nbox, const float* pres, const float* depth, int ld_pd, void* rec, int B, int HW, int I, int P, int ac, hipStream_t s);
int render_fwd_mma(const void* S16, int ld_s, const void* rec, const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int I, int P, int ac, hipStream_t s);
int render_bwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux, const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int C, int I, int P, int ac, float obj_scale, float alpha_scale, int g_bf16, int s_bf16, const void* rec, hipStream_t s);
int render_prep_supported(int HW, int I, int P, int ac);
int render_fwd_c(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x, float* recon,
                 float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int ac, hipStream_t s);
int render_bwd_c(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
                 const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int C, int I, int P,
                 int ac, float obj_scale, float alpha_scale, hipStream_t s);
int loss_count_kl(const CellLayout& L, const CellBufs& P, float prior_prob, float* klp, hipStream_t s);
int loss_gauss_kl_blocks(const CellLayout& L);
int loss_gauss_kl(const CellLayout& L, const CellBufs& P, const CellHyper& H, float* partial, hipStream_t s);
int loss_finalize(const float* bce_partial, int n_bce, const float* kl_partial, int n_kl, const float* klp, int B, float kl_scale, float beta, float* loss_out, const int* failed, int* status, int* status_host, hipStream_t s);

#define TRY(expr)                      \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != SPAIR_OK) return rc__; \
    } while (0)


// ---------------------------------------------------------------------------------------------
// opt-in instrumentation (bench.py): HIP events around regions / single kernels ON THE CALLER'S
// STREAM.  Off by default; the step itself stays stateless.
// ---------------------------------------------------------------------------------------------
#define SP_PROF_SLOTS 24
#define SP_PROF_POOL 8192
struct ProfState {
    int enabled;
    int n_events;                 // created
    unsigned long long mask = ~0ull;   // slots recorded (spair_prof_select)
    int used;                     // event pairs used
    hipEvent_t ev[2 * SP_PROF_POOL];
    int slot[SP_PROF_POOL];
};
static ProfState g_prof;            // opt-in, process-wide, guarded by g_prof_mu (bench.py only; off by default)
static std::mutex g_prof_mu;

static inline int prof_begin(int slot, hipStream_t s) {
    if (slot < 0 || !g_prof.enabled) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!((g_prof.mask >> slot) & 1ull) || g_prof.used >= SP_PROF_POOL) return -1;
    const int i = g_prof.used++;
    g_prof.slot[i] = slot;
    hipEventRecord(g_prof.ev[2 * i], s);
    return i;
}
static inline void prof_end(int i, hipStream_t s) {
    if (i >= 0) hipEventRecord(g_prof.ev[2 * i + 1], s);
}
struct ProfScope {
    int i; hipStream_t s;
    ProfScope(int slot, hipStream_t st) : i(prof_begin(slot, st)), s(st) {}
    ~ProfScope() { prof_end(i, s); }
};

extern "C" int spair_prof_enable(int enable) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (enable && g_prof.n_events == 0) {
        for (int i = 0; i < 2 * SP_PROF_POOL; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return SPAIR_ERR_LAUNCH;
        g_prof.n_events = 2 * SP_PROF_POOL;
    }
    g_prof.enabled = enable ? 1 : 0;
    if (enable == 1) g_prof.used = 0;      // 0 (stop) and 2 (resume) keep what was recorded: sampling every n-th step of a timed region
    return SPAIR_OK;
}
// Restricts the recorded regions to the slots whose bit is set (default: all).
extern "C" int spair_prof_select(unsigned long long mask) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof.mask = mask;
    return SPAIR_OK;
}
// Synchronises on the recorded events (call only outside the timed region); ms[slot] += elapsed, counts[slot] += 1.
extern "C" int spair_prof_read(float* ms, int* counts, int nslots) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (int i = 0; i < g_prof.used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return SPAIR_ERR_LAUNCH;
        if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return SPAIR_ERR_LAUNCH;
        const int sl = g_prof.slot[i];
        if (sl >= 0 && sl < nslots) { ms[sl] += t; counts[sl] += 1; }
    }
    g_prof.used = 0;
    return SPAIR_OK;
}
enum { PS_PREP = 0, PS_BACKBONE_FWD, PS_CELLS_FWD, PS_DECODER_FWD, PS_COUNT_KL, PS_RENDER_FWD, PS_LOSS, PS_RENDER_BWD, PS_DECODER_BWD,
       PS_CELLS_BWD, PS_CELLS_WGRAD, PS_BACKBONE_BWD, PS_CONV1_FWD, PS_DEC2_FWD, PS_STN_FWD, PS_ADAM, PS_DEC2_WGRAD, PS_DEC2_DGRAD };

static int validate(const SpairDims& d) {
    if (d.B <= 0 || d.I <= 0 || d.G <= 0 || d.P <= 0 || d.A <= 0 || d.F <= 0 || d.NP <= 0) return SPAIR_ERR_SHAPE;
    if (d.C < 1 || d.C > 3) return SPAIR_ERR_UNSUPPORTED;  // colour channels: 1 (every tuned kernel) .. 3 (generic-channel renderer, render_c.hip)
    if (d.C != 1 && d.obj_conv) return SPAIR_ERR_UNSUPPORTED;      // colour images: the per-wavefront step (either dtype), MLP object nets
    if (d.n_conv < 1 || d.n_conv > SP_MAX_CONV) return SPAIR_ERR_SHAPE;
    if (d.dtype != SPAIR_F32 && d.dtype != SPAIR_BF16) return SPAIR_ERR_DTYPE;
    if ((d.F & 3) || (d.NP & 3)) return SPAIR_ERR_ALIGN;
    int h = d.I + d.pad_pre + d.pad_post;
    for (int i = 0; i < d.n_conv; ++i) {
        if (d.conv_k[i] < 1 || d.conv_s[i] < 1 || (d.conv_c[i] & 7)) return SPAIR_ERR_SHAPE;
        if (i > 0 && d.conv_k[i] > 1 && (d.conv_k[i] % d.conv_s[i] != 0 || d.conv_s[i] > 2)) return SPAIR_ERR_UNSUPPORTED;
        if (i > 0 && d.conv_k[i] == 1 && d.conv_s[i] != 1) return SPAIR_ERR_UNSUPPORTED;
        h = (h - d.conv_k[i]) / d.conv_s[i] + 1;
    }
    if (h != d.G) return SPAIR_ERR_SHAPE;
    if (d.G * d.G + 1 > 1025) return SPAIR_ERR_UNSUPPORTED;
    if (d.A + 5 > 64) return SPAIR_ERR_UNSUPPORTED;        // k_gauss_kl (loss.hip): one lane per latent element of a cell (A attributes + 4 box + 1 depth)
    if (d.lookback < 0 || d.lookback > 3) return SPAIR_ERR_UNSUPPORTED;
    if (d.lookback > 1 && d.G > 32) return SPAIR_ERR_UNSUPPORTED;
    if (d.obj_conv) {   // convolutional object encoder / decoder variant: per-wavefront launches, the convolutions themselves in fp32 (objconv.hip)
        if (d.oc_n < 1 || d.oc_n > 4) return SPAIR_ERR_SHAPE;
        int hh = d.P, ci = d.C;
        for (int i = 0; i < d.oc_n; ++i) {
            if (d.oc_k[i] < 1 || d.oc_s[i] < 1 || d.oc_c[i] < 1 || d.oc_k[i] > hh) return SPAIR_ERR_SHAPE;
            if ((long long)ci * d.oc_c[i] * d.oc_k[i] * d.oc_k[i] > 256 * 36 || d.oc_c[i] > 256) return SPAIR_ERR_UNSUPPORTED;
            const int ho = (hh - d.oc_k[i]) / d.oc_s[i] + 1;
            hh = ho; ci = d.oc_c[i];
        }
        if (((ci * hh * hh) & 7) || ((d.P * d.P * (d.C + 1)) & 7)) return SPAIR_ERR_ALIGN;       // rows handed to the GEMM / column-sum kernels
        if ((long long)(d.C + 1) * d.oc_c[0] * d.oc_k[0] * d.oc_k[0] > 256 * 36) return SPAIR_ERR_UNSUPPORTED;
    }
    return SPAIR_OK;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
struct Ws {
    int *cell_h, *cell_w, *cidx, *nbr, *cons, *diag_start;
    void* conv_wf[SP_MAX_CONV + 1];
    void* conv_wd[SP_MAX_CONV + 1][4];
    void* lin_wf[LIN_COUNT];     // heads: BOXH1 / ZH1 slots hold the concatenated [pass | lat] matrix
    void* lin_wt[LIN_COUNT];
    float *bias_boxh, *bias_zh;
    void* chain_w[CW_COUNT];      // fragment-packed weights for the fused chain (bf16)
    void* chain_wlo[3];           // low parts of the box network's packs (split-bf16 forward)
    void* chain_wt[CW_COUNT];     // ... and their data-gradient packs
    void* dec_stream;             // fused decoder forward: fragment stream of its three weight matrices (bf16 mode)
    float* xpad;
    float *act[SP_MAX_CONV + 1], *dact[SP_MAX_CONV + 1];
    void* dLog16;                     // bf16 copy of dLog (colour images in the bf16 step only)
    unsigned char* act0_bits;         // sign bits of act[0] (one byte per pixel and 8 channels): conv_1's data-gradient gate, written by the stem kernel
    unsigned char* act_bits[SP_MAX_CONV + 1];      // the same for the outputs of the patch-resident strided convolutions (gates of the next layer's dgrad)
    float *feat, *dfeat;
    CellBufs cb;
    float *Za, *Hd1, *Hd2, *S, *dLog, *dHd2, *dHd1;     // Hd*, dLog, dHd*: bf16 in bf16 mode
    void *Za16, *dfeat16;                               // bf16 copies of the decoder input / d feat (bf16 mode)
    float *tn_part, *tn_part2;                          // split-K partial tiles of the weight-gradient GEMMs (caller's / helper stream)
    float *aux, *bce_partial, *kl_partial, *klp, *gedge_part;
    void* rrec;
    // convolutional object encoder / decoder variant (fp32): layer outputs and their gradients, N rows each
    float *oc_ea[4], *oc_dea[4];      // encoder conv i (post-ReLU; the last one in (C,H,W) order = the Linear's input), d of the same
    float *oc_dh, *oc_ddh;            // decoder Linear output (C,H,W) and its gradient
    float *oc_da[4], *oc_dda[4];      // decoder transposed conv i < n-1 (post-ReLU, NHWC), d of the same (the last one writes the sprites)
    int* chain_sync;                  // band split of the fused chain (chain.h): start tickets, time-out word, per-(sample, band) counters
    float *bnd_rec, *bnd_grad;        // ... and the boundary rows handed from band to band
    unsigned long long* stamps;
    int ld_feat, ld_s;
    size_t total;
};

struct Carver {
    char* base;
    size_t off;
    template <class T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
    void* take_bytes(size_t n) { return take<char>(n); }
};

static Ws carve(const SpairDims& d, void* base) {
    Ws w;
    memset(&w, 0, sizeof(w));
    const CellLayout L = make_cell_layout(d);
    const ParamLayout PL = make_param_layout(d);
    Carver c{reinterpret_cast<char*>(base), 0};
    const size_t es = d.dtype == SPAIR_BF16 ? 2 : 4;
    const size_t N = (size_t)L.N;
    w.cell_h = c.take<int>(L.HW); w.cell_w = c.take<int>(L.HW); w.cidx = c.take<int>(L.HW);
    w.nbr = c.take<int>(L.NB * L.HW); w.cons = c.take<int>(L.NB * L.HW); w.diag_start = c.take<int>((L.LB + 2) * L.G + 2);
    // prepared weights
    for (int i = 1; i < PL.n_conv; ++i) {
        const ConvSpec& cs = PL.conv[i];
        const size_t K = (size_t)cs.k * cs.k * cs.cin;
        w.conv_wf[i] = c.take_bytes((size_t)cs.cout * round_up((int)K, 8) * es);
        if (cs.k == 1) w.conv_wd[i][0] = c.take_bytes((size_t)cs.cin * round_up(cs.cout, 8) * es);
        else for (int q = 0; q < cs.s * cs.s; ++q) {
            const int T = cs.k / cs.s;
            w.conv_wd[i][q] = c.take_bytes((size_t)cs.cin * T * T * cs.cout * es);
        }
    }
    auto lin_alloc = [&](int id, int rows_total) {
        w.lin_wf[id] = c.take_bytes((size_t)rows_total * round_up(PL.lin[id].in, 8) * es);
        w.lin_wt[id] = c.take_bytes((size_t)PL.lin[id].in * round_up(rows_total, 8) * es);
    };
    for (int id = 0; id < LIN_COUNT; ++id) {
        if (id == LIN_BOXH0 || id == LIN_ZH0) continue;
        int rows = PL.lin[id].out;
        if (id == LIN_BOXH1) rows += PL.lin[LIN_BOXH0].out;
        if (id == LIN_ZH1) rows += PL.lin[LIN_ZH0].out;
        lin_alloc(id, rows);
    }
    w.bias_boxh = c.take<float>(L.NP + 8);
    w.bias_zh = c.take<float>(L.NP + 8);
    if (chain_fwd_supported(d)) {
        const int nt[CW_COUNT] = {7, 7, 7, 16, 8, 7, 7, 7, 7, 7, 7, 1};
        const int kt[CW_COUNT] = {11, 4, 4, 25, 8, 4, 16, 4, 4, 16, 4, 4};
        for (int i = 0; i < CW_COUNT; ++i) w.chain_w[i] = c.take_bytes((size_t)nt[i] * kt[i] * 1024);
        for (int i = 0; i < 3; ++i) w.chain_wlo[i] = c.take_bytes((size_t)nt[i] * kt[i] * 1024);
        const int ntb[CW_COUNT] = {21, 7, 7, 49, 16, 8, 30, 7, 7, 30, 7, 0};
        const int ktb[CW_COUNT] = {4, 4, 4, 8, 4, 4, 4, 4, 4, 4, 4, 0};
        for (int i = 0; i < CW_COUNT; ++i) w.chain_wt[i] = ntb[i] ? c.take_bytes((size_t)ntb[i] * ktb[i] * 1024) : nullptr;
    }
    w.dec_stream = d.dtype == SPAIR_BF16 ? c.take_bytes(dec_fused_stream_bytes(d.P * d.P * (d.C + 1))) : nullptr;
    // backbone
    const int Ip = d.I + d.pad_pre + d.pad_post;
    w.xpad = c.take<float>((size_t)d.B * Ip * Ip * d.C);
    for (int i = 0; i < d.n_conv; ++i) {
        const ConvSpec& cs = PL.conv[i];
        const size_t n = (size_t)d.B * cs.hout * cs.hout * cs.cout;
        w.act[i] = reinterpret_cast<float*>(c.take_bytes(n * es));      // NHWC, bf16 in bf16 mode
        w.dact[i] = reinterpret_cast<float*>(c.take_bytes(n * es));
    }
    {
        const ConvSpec& c0 = PL.conv[0];
        w.act0_bits = misc_conv0_writes_mask(d.B, c0.hin, d.C, c0.k, c0.s, c0.cout, d.dtype == SPAIR_BF16)
                          ? c.take<unsigned char>((size_t)d.B * c0.hout * c0.hout * 16) : nullptr;
    }
    for (int i = 1; i < d.n_conv; ++i) {
        const ConvSpec& cs = PL.conv[i];
        w.act_bits[i] = (d.dtype == SPAIR_BF16 && cs.k == 4 && cs.s == 2 && cs.cin == 128 && cs.cout == 128)
                            ? c.take<unsigned char>((size_t)d.B * cs.hout * cs.hout * 16) : nullptr;
    }
    w.ld_feat = round_up(d.F, 8);
    w.feat = c.take<float>(N * w.ld_feat);
    w.dfeat = c.take<float>(N * w.ld_feat);
    // per-cell chain
    CellBufs& b = w.cb;
    b.cell_h = w.cell_h; b.cell_w = w.cell_w; b.cidx = w.cidx; b.nbr = w.nbr; b.cons = w.cons; b.diag_start = w.diag_start;
    b.feat = w.feat; b.ld_feat = w.ld_feat; b.dfeat = w.dfeat;
    b.Xb = c.take<float>(N * L.ld_xb); b.Hb1 = c.take<float>(N * SP_LDH); b.Hb2 = c.take<float>(N * SP_LDH); b.Ob = c.take<float>(N * L.ld_ob);
    b.glimpse = c.take<float>(N * L.ld_gl); b.He1 = c.take<float>(N * SP_ENC_H1); b.He2 = c.take<float>(N * SP_ENC_H2); b.Oe = c.take<float>(N * L.ld_oe);
    b.Xz = c.take<float>(N * L.ld_x); b.Hz1 = c.take<float>(N * SP_LDH); b.Hz2 = c.take<float>(N * SP_LDH); b.Oz = c.take<float>(N * L.ld_oz);
    b.Xo = c.take<float>(N * L.ld_x); b.Ho1 = c.take<float>(N * SP_LDH); b.Ho2 = c.take<float>(N * SP_LDH); b.Oo = c.take<float>(N * L.ld_oo);
    b.rec = c.take<float>(N * L.ld_rec); b.sd_attr = c.take<float>(N * L.ld_rec); b.nbox = c.take<float>(N * 4); b.stat = c.take<float>(N * SP_LDSTAT);
    b.dXb = c.take<float>(N * L.ld_xb); b.dHb1 = c.take<float>(N * SP_LDH); b.dHb2 = c.take<float>(N * SP_LDH); b.dOb = c.take<float>(N * L.ld_ob);
    b.dGl = c.take<float>(N * L.ld_gl); b.dHe1 = c.take<float>(N * SP_ENC_H1); b.dHe2 = c.take<float>(N * SP_ENC_H2); b.dOe = c.take<float>(N * L.ld_oe);
    b.dXz = c.take<float>(N * L.ld_x); b.dHz1 = c.take<float>(N * SP_LDH); b.dHz2 = c.take<float>(N * SP_LDH); b.dOz = c.take<float>(N * L.ld_oz);
    b.dXo = c.take<float>(N * L.ld_x); b.dHo1 = c.take<float>(N * SP_LDH); b.dHo2 = c.take<float>(N * SP_LDH); b.dOo = c.take<float>(N * L.ld_oo);
    b.grec = c.take<float>(N * L.ld_rec); b.g_nbox_stn = c.take<float>(N * 4);
    b.g_nbox_r = c.take<float>(N * 4); b.g_pres_r = c.take<float>(N); b.g_depth_r = c.take<float>(N); b.g_attr_r = c.take<float>(N * L.ld_rec);
    // decoder / renderer / loss
    w.ld_s = round_up(d.P * d.P * (d.C + 1), 8);
    w.Za = c.take<float>(N * L.ld_rec);
    b.Za = w.Za;
    b.gxy = chain_fwd_supported(d) ? c.take<unsigned int>(N * L.ld_gl) : nullptr;
    const int nbands = chain_fwd_supported(d) ? chain_bands(d) : 1;
    b.mbits = chain_fwd_supported(d) ? c.take<unsigned long long>((size_t)d.B * nbands * (3 * d.G - 2) * 66 * 4) : nullptr;
    w.Hd1 = reinterpret_cast<float*>(c.take_bytes(N * SP_DEC_H1 * es)); w.Hd2 = reinterpret_cast<float*>(c.take_bytes(N * SP_DEC_H2 * es));
    w.S = c.take<float>(N * w.ld_s);
    // (the conv decoder takes fp32 sprite gradients in both modes; the generic-channel renderer of colour images writes fp32 ones, which the
    //  bf16 step's decoder backward reads through a bf16 copy)
    w.dLog = reinterpret_cast<float*>(c.take_bytes(N * w.ld_s * ((d.obj_conv || d.C != 1) ? 4 : es)));
    w.dLog16 = (d.dtype == SPAIR_BF16 && d.C != 1) ? c.take_bytes(N * w.ld_s * 2) : nullptr;
    w.dHd2 = reinterpret_cast<float*>(c.take_bytes(N * SP_DEC_H2 * es)); w.dHd1 = reinterpret_cast<float*>(c.take_bytes(N * SP_DEC_H1 * es));
    w.Za16 = c.take_bytes(N * L.ld_rec * 2); w.dfeat16 = c.take_bytes(N * w.ld_feat * 2);
    b.Za16 = w.Za16; b.dfeat16 = w.dfeat16;
    w.tn_part = reinterpret_cast<float*>(c.take_bytes((size_t)SPAIR_TN_PART_FLOATS * 4));
    w.tn_part2 = reinterpret_cast<float*>(c.take_bytes((size_t)SPAIR_TN_PART_FLOATS * 4));
    w.aux = c.take<float>((size_t)d.B * d.C * d.I * d.I * 2);     // float2 per pixel and colour channel: (dBCE/dpre / D, pre)
    w.bce_partial = c.take<float>(render_num_blocks(d.B, d.I));
    w.rrec = c.take_bytes((size_t)N * 64);                  // the renderer's per-object records (render3.hip)
    for (int i = 0; i < PL.oc_n; ++i) {
        const ConvSpec& e = PL.oc_enc[i];
        const size_t n = N * e.hout * e.hout * e.cout;
        w.oc_ea[i] = c.take<float>(n); w.oc_dea[i] = c.take<float>(n);
        if (i + 1 < PL.oc_n) {
            const ConvSpec& t = PL.oc_dec[i];
            const size_t m = N * t.hout * t.hout * t.cout;
            w.oc_da[i] = c.take<float>(m); w.oc_dda[i] = c.take<float>(m);
        }
    }
    if (PL.oc_n) { w.oc_dh = c.take<float>(N * PL.oc_flat); w.oc_ddh = c.take<float>(N * PL.oc_flat); }
    w.kl_partial = c.take<float>((size_t)loss_gauss_kl_blocks(L) * 6);
    w.klp = c.take<float>(d.B);
    w.gedge_part = c.take<float>((size_t)d.B * nbands * 4 * L.REC);
    w.chain_sync = nbands > 1 ? c.take<int>((size_t)CHAIN_SYNC_WORDS(d.B, nbands)) : nullptr;
    w.bnd_rec = nbands > 1 ? c.take<float>((size_t)d.B * nbands * d.G * L.REC) : nullptr;
    w.bnd_grad = nbands > 1 ? c.take<float>((size_t)d.B * nbands * d.G * 3 * L.REC) : nullptr;
    w.stamps = c.take<unsigned long long>(4096);
    w.total = (c.off + 255) & ~(size_t)255;
    return w;
}

extern "C" int64_t spair_workspace_bytes(const SpairDims* d) {
    if (!d || validate(*d) != SPAIR_OK) return -1;
    return (int64_t)carve(*d, nullptr).total;
}

// ---------------------------------------------------------------------------------------------
// parameter naming (reference state_dict keys)
// ---------------------------------------------------------------------------------------------
struct PInfo { char name[96]; int64_t off; int64_t shape[4]; int ndim; };

static std::vector<PInfo> param_infos(const SpairDims& d) {
    const ParamLayout P = make_param_layout(d);
    const CellLayout L = make_cell_layout(d);
    std::vector<PInfo> v;
    auto add = [&](const char* nm, int64_t off, int nd, int64_t s0, int64_t s1 = 1, int64_t s2 = 1, int64_t s3 = 1) {
        PInfo p;
        snprintf(p.name, sizeof(p.name), "%s", nm);
        p.off = off; p.ndim = nd; p.shape[0] = s0; p.shape[1] = s1; p.shape[2] = s2; p.shape[3] = s3;
        v.push_back(p);
    };
    char buf[96];
    add("virtual_edge_element", P.edge, 1, L.REC);
    for (int i = 0; i < d.n_conv; ++i) {
        const ConvSpec& c = P.conv[i];
        snprintf(buf, sizeof(buf), "backbone.net.conv_%d.weight", i); add(buf, c.w, 4, c.cout, c.cin, c.k, c.k);
        snprintf(buf, sizeof(buf), "backbone.net.conv_%d.bias", i); add(buf, c.b, 1, c.cout);
    }
    {
        const ConvSpec& c = P.conv[d.n_conv];
        add("backbone.net.conv_out.weight", c.w, 4, c.cout, c.cin, 1, 1);
        add("backbone.net.conv_out.bias", c.b, 1, c.cout);
    }
    auto lin = [&](const char* nm, int id) {
        snprintf(buf, sizeof(buf), "%s.weight", nm); add(buf, P.lin[id].w, 2, P.lin[id].out, P.lin[id].in);
        snprintf(buf, sizeof(buf), "%s.bias", nm); add(buf, P.lin[id].b, 1, P.lin[id].out);
    };
    lin("box_network.body.dense0", LIN_BOX0); lin("box_network.body.dense1", LIN_BOX1);
    lin("box_network.output_layers.0", LIN_BOXH0); lin("box_network.output_layers.1", LIN_BOXH1);
    if (P.oc_n) {
        for (int i = 0; i < P.oc_n; ++i) {
            const ConvSpec& c = P.oc_enc[i];
            snprintf(buf, sizeof(buf), "object_encoder.conv.conv_%d.weight", i); add(buf, c.w, 4, c.cout, c.cin, c.k, c.k);
            snprintf(buf, sizeof(buf), "object_encoder.conv.conv_%d.bias", i); add(buf, c.b, 1, c.cout);
        }
        lin("object_encoder.out", LIN_ENC2);
    } else {
    lin("object_encoder.dense0", LIN_ENC0); lin("object_encoder.dense1", LIN_ENC1); lin("object_encoder.out", LIN_ENC2);
    }
    lin("z_network.body.dense0", LIN_Z0); lin("z_network.body.dense1", LIN_Z1);
    lin("z_network.output_layers.0", LIN_ZH0); lin("z_network.output_layers.1", LIN_ZH1);
    lin("obj_network.dense0", LIN_OBJ0); lin("obj_network.dense1", LIN_OBJ1); lin("obj_network.out", LIN_OBJ2);
    if (P.oc_n) {
        lin("object_decoder.inp", LIN_DEC0);
        for (int i = 0; i < P.oc_n; ++i) {
            const ConvSpec& c = P.oc_dec[i];
            snprintf(buf, sizeof(buf), "object_decoder.conv.conv_transposed_%d.weight", i); add(buf, c.w, 4, c.cin, c.cout, c.k, c.k);
            snprintf(buf, sizeof(buf), "object_decoder.conv.conv_transposed_%d.bias", i); add(buf, c.b, 1, c.cout);
        }
    } else {
    lin("object_decoder.dense0", LIN_DEC0); lin("object_decoder.dense1", LIN_DEC1); lin("object_decoder.out", LIN_DEC2);
    }
    const int ad = 4 + d.A + 1;
    add("attn.gamma", P.attn_gamma, 1, 1);
    add("attn.query_conv.weight", P.attn_q_w, 4, ad / 8, ad, 1, 1); add("attn.query_conv.bias", P.attn_q_b, 1, ad / 8);
    add("attn.key_conv.weight", P.attn_k_w, 4, ad / 8, ad, 1, 1); add("attn.key_conv.bias", P.attn_k_b, 1, ad / 8);
    add("attn.value_conv.weight", P.attn_v_w, 4, ad, ad, 1, 1); add("attn.value_conv.bias", P.attn_v_b, 1, ad);
    return v;
}

extern "C" int spair_param_count(const SpairDims* d) { return d ? (int)param_infos(*d).size() : -1; }
extern "C" int64_t spair_param_total(const SpairDims* d) { return d ? make_param_layout(*d).total : -1; }
extern "C" int spair_param_info(const SpairDims* d, int idx, char* name, int name_cap, int64_t* offset, int64_t* shape4, int* ndim) {
    if (!d) return SPAIR_ERR_SHAPE;
    const std::vector<PInfo> v = param_infos(*d);
    if (idx < 0 || idx >= (int)v.size()) return SPAIR_ERR_SHAPE;
    snprintf(name, name_cap, "%s", v[idx].name);
    *offset = v[idx].off;
    for (int i = 0; i < 4; ++i) shape4[i] = v[idx].shape[i];
    *ndim = v[idx].ndim;
    return SPAIR_OK;
}

// ---------------------------------------------------------------------------------------------
// step context
// ---------------------------------------------------------------------------------------------
struct Ctx {
    SpairDims d;
    SpairStep st;
    CellLayout L;
    ParamLayout PL;
    Ws w;
    CellHyper H;
    const float* params;
    const float* x;
    hipStream_t s;
    int T;                 // number of wavefront diagonals
    int use_chain;         // fused persistent per-cell kernels (bf16, reference network sizes)
    int use_dec_fused;     // the decoder forward as one activation-stationary kernel (bf16; SpairStep.flags bit 4 turns it off)
    float* tn_scratch = nullptr;   // split-K scratch override while work is being issued on the helper stream
    std::vector<int> dstart;
};

// ---- helper stream --------------------------------------------------------------------------------
// Stages that do not depend on each other are issued on a second HIP stream (its own hardware queue) so that
// latency-bound kernels (the per-sample chain: one workgroup per CU, the count-KL scan: one wave per sample) share the
// chip with throughput kernels (weight-gradient GEMMs, decoder, renderer).  Fork/join is by events only, so the
// caller's stream still sees the whole step in order and the step stays capturable in a hipGraph.
// SpairStep.flags bit 2 disables it (everything on the caller's stream).
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[6];
    std::mutex enq_mu;      // held for the whole enqueue of one spair_forward / spair_backward call that uses this helper stream
};
// One helper stream + its fork/join events PER DEVICE, created once under a lock (first use, or spair_init() ahead of a hipGraph
// capture -- stream / event creation is not capturable).  They carry no data between calls: every call forks from and joins back
// into the caller's stream.  The six events are shared by every call on the device, so a call holds enq_mu from its first fork to its
// last join: hipStreamWaitEvent binds to the record that precedes it at CALL time, hence two host threads driving two caller streams
// of one device can never wait on each other's records (they only share the helper queue's ordering) -- without the lock thread A
// could wait on the record thread B had just made on ITS stream and start its helper-stream weight gradients before its own producers.
#define SP_MAX_DEVICES 64
static SideStream g_side[SP_MAX_DEVICES];
static std::mutex g_side_mu;
static int side_stream(SideStream*& out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SP_MAX_DEVICES) return SPAIR_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_side_mu);
    SideStream& sd = g_side[dev];
    if (sd.s == nullptr) {
        // lowest priority: what runs here (weight gradients, KL terms, preparation) has slack, the caller's stream carries the
        // dependent chain -- when both have workgroups to place, the chain's go first
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = 0;
        hipStream_t st = nullptr;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_least) != hipSuccess) return SPAIR_ERR_LAUNCH;
        for (int i = 0; i < 6; ++i)
            if (hipEventCreateWithFlags(&sd.ev[i], hipEventDisableTiming) != hipSuccess) return SPAIR_ERR_LAUNCH;
        sd.s = st;
    }
    out = &sd;
    return SPAIR_OK;
}
// Creates the current device's helper stream ahead of time (optional; spair_forward / spair_backward do it on first use).
extern "C" int spair_init(void) {
    SideStream* sd = nullptr;
    return side_stream(sd);
}
// everything enqueued on `to` after this call runs after everything enqueued on `from` before it
static int stream_link(hipStream_t from, hipStream_t to, hipEvent_t e) {
    if (hipEventRecord(e, from) != hipSuccess) return SPAIR_ERR_LAUNCH;
    if (hipStreamWaitEvent(to, e, 0) != hipSuccess) return SPAIR_ERR_LAUNCH;
    return SPAIR_OK;
}

static void fill_diag(Ctx& c) {
    const int G = c.d.G;
    const int S = c.L.LB + 1;             // dependency wavefronts t = (N_LOOKBACK + 1) h + w (cells.hip, k_init_tables)
    c.T = S * (G - 1) + G;
    c.dstart.assign(c.T + 1, 0);
    int n = 0;
    for (int t = 0; t < c.T; ++t) {
        c.dstart[t] = n;
        for (int h = 0; h < G; ++h) {
            const int w = t - S * h;
            if (w >= 0 && w < G) ++n;
        }
    }
    c.dstart[c.T] = n;
}

static int make_ctx(Ctx& c, const SpairDims* d, const SpairStep* st, const float* params, const float* x, const float* eps_box,
                    const float* eps_attr, const float* eps_depth, const float* u_pres, void* workspace, void* stream) {
    if (!d || !st || !params || !x || !workspace) return SPAIR_ERR_SHAPE;
    TRY(validate(*d));
    c.d = *d; c.st = *st;
    c.L = make_cell_layout(*d);
    c.PL = make_param_layout(*d);
    c.w = carve(*d, workspace);
    c.params = params; c.x = x; c.s = (hipStream_t)stream;
    CellHyper& H = c.H;
    H.wheel = st->wheel; H.kl_scale = st->kl_scale * d->vae_beta; H.img = (float)d->I; H.anchor = d->anchor;
    H.cell_over_img = (float)((double)d->cell_px / (double)d->I);
    H.max_yx = d->max_yx; H.min_yx = d->min_yx; H.max_hw = d->max_hw; H.min_hw = d->min_hw;
    H.range_yx = d->max_yx - d->min_yx; H.range_hw = d->max_hw - d->min_hw;
    for (int i = 0; i < 6; ++i) { H.prior_mean[i] = d->prior_mean[i]; H.prior_std[i] = d->prior_std[i]; }
    H.count_prior_prob = st->count_prior_prob;
    c.w.cb.edge = params + c.PL.edge;
    c.w.cb.eps_box = eps_box; c.w.cb.eps_attr = eps_attr; c.w.cb.eps_depth = eps_depth; c.w.cb.u_pres = u_pres;
    fill_diag(c);
    c.use_chain = chain_fwd_supported(*d) && !(st->flags & 1) && !d->obj_conv;
    c.use_dec_fused = !d->obj_conv && d->dtype == SPAIR_BF16 && !(st->flags & 16) && c.PL.lin[LIN_DEC0].out == SP_DEC_H1 && c.PL.lin[LIN_DEC1].out == SP_DEC_H2 &&
                      dec_fused_supported(d->A, d->P * d->P * (d->C + 1), c.L.ld_rec, c.L.N, c.w.ld_s) && d->C == 1;
    return SPAIR_OK;
}

// ---- GEMM helpers ---------------------------------------------------------------------------------
static int nt(Ctx& c, const float* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, const float* bias,
              const float* mask, int ldmask, int relu) {
    GemmNT g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.mask = mask; g.ldmask = ldmask; g.relu = relu;
    return spair_gemm_nt_impl(g, false, c.d.dtype, c.s);
}
static int tn(Ctx& c, const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc, int R, float* colsum = nullptr) {
    GemmTN g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = round_up(M, 4); g.N = round_up(N, 4); g.Mstore = M; g.Nstore = N; g.R = R; g.colsum_out = colsum;
    // (split-K partial tiles do not pay here: N-contiguous atomics on 1-3 tiles are cheaper than the extra pass)
    return spair_gemm_tn_impl(g, false, c.d.dtype, c.s);
}
// bf16-stored operand GEMMs (gemm16.hip)
// strided k x k convs of the bf16 step keep their forward weights in tap-parity K order (gemm.h, GemmNT::ktab)
static bool conv_kperm(const Ctx& c, const ConvSpec& cs) {
    return c.d.dtype == SPAIR_BF16 && cs.k > 1 && cs.s > 1 && cs.k % cs.s == 0 && cs.cin % 64 == 0 && cs.k * cs.k * cs.cin / 64 <= 64;
}
static void fill_ktab(GemmNT& g, const ConvSpec& cs) {
    const int T = cs.k / cs.s, TT = T * T, nh = cs.cin / 64;
    g.n_ktab = cs.k * cs.k * nh;
    for (int blk = 0; blk < g.n_ktab; ++blk) {
        const int tq = blk % TT, rr = blk / TT, h = rr % nh, cls = rr / nh;
        const int ky = cls / cs.s + cs.s * (tq / T), kx = cls % cs.s + cs.s * (tq % T);
        g.ktab[blk] = ((unsigned)ky << 24) | ((unsigned)kx << 16) | (unsigned)(h * 64);
    }
}
static int nt16(Ctx& c, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_bf16, int M, int N, int K, const float* bias,
                const void* mask, int ldmask, int relu, const ConvDesc* conv = nullptr, const RowMap* cmap = nullptr,
                const ConvSpec* fwd_cs = nullptr) {
    GemmNT g;
    memset(&g, 0, sizeof(g));
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.B = B; g.ldb = ldb; g.C = reinterpret_cast<float*>(C); g.ldc = ldc; g.c_bf16 = c_bf16;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.mask = reinterpret_cast<const float*>(mask); g.ldmask = ldmask; g.mask_bf16 = 1; g.relu = relu;
    if (conv) g.conv = *conv;
    if (cmap) { g.cmap = *cmap; g.use_cmap = 1; }
    if (fwd_cs && conv_kperm(c, *fwd_cs)) fill_ktab(g, *fwd_cs);       // forward conv: B (conv_wf) is stored in tap-parity K order
    return spair_gemm_nt16_impl(g, conv != nullptr, c.s);
}
static int tn16(Ctx& c, const void* A, int lda, int M, const void* B, int ldb, int N, bool b_bf16, float* C, int ldc, int R, float* colsum,
                const ConvDesc* conv = nullptr, int cw_cin = 0, int cw_taps = 0) {
    GemmTN g;
    memset(&g, 0, sizeof(g));
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.B = reinterpret_cast<const float*>(B); g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = round_up(M, 8); g.N = round_up(N, b_bf16 ? 8 : 4); g.Mstore = M; g.Nstore = N; g.R = R; g.colsum_out = colsum;
    g.cw_cin = cw_cin; g.cw_taps = cw_taps;
    g.part = c.tn_scratch ? c.tn_scratch : c.w.tn_part; g.part_cap = SPAIR_TN_PART_FLOATS;
    if (conv) g.conv = *conv;
    return spair_gemm_tn16_impl(g, conv != nullptr, b_bf16, c.s);
}
static const void* bptr(const void* base, size_t elem_off, int dtype) {
    return reinterpret_cast<const char*>(base) + elem_off * (dtype == SPAIR_BF16 ? 2 : 4);
}

// ---- weight preparation ----------------------------------------------------------------------------
// part 0: the backbone's conv weights (what conv_1 waits for); part 1: everything else (first needed by the per-cell chain)
static int prep_weights(Ctx& c, bool need_dgrad, int part) {
    std::vector<PrepEntry> es;
    const int bf = c.d.dtype == SPAIR_BF16;
    auto push = [&](const float* src, void* dst, int rows, int cols, int ld, int mode, int bf16, int cin = 0, int cout = 0, int k = 0,
                    int py = 0, int px = 0, int T = 0, int s = 0) {
        PrepEntry e;
        memset(&e, 0, sizeof(e));
        e.src = src; e.dst = dst; e.rows = rows; e.cols = cols; e.ld = ld; e.mode = mode; e.bf16 = bf16;
        e.cin = cin; e.cout = cout; e.k = k; e.py = py; e.px = px; e.T = T; e.s = s;
        es.push_back(e);
    };
    for (int i = 1; part == 0 && i < c.PL.n_conv; ++i) {
        const ConvSpec& cs = c.PL.conv[i];
        const float* w = c.params + cs.w;
        const int K = cs.k * cs.k * cs.cin;
        if (cs.k == 1) {
            push(w, c.w.conv_wf[i], cs.cout, cs.cin, round_up(K, 8), 0, bf);
            if (need_dgrad) push(w, c.w.conv_wd[i][0], cs.cin, cs.cout, round_up(cs.cout, 8), 1, bf);
        } else {
            const bool perm = conv_kperm(c, cs);
            push(w, c.w.conv_wf[i], cs.cout, K, round_up(K, 8), 2, bf, cs.cin, cs.cout, cs.k, 0, 0, perm ? cs.k / cs.s : 0, perm ? cs.s : 0);
            if (need_dgrad) {
                const int T = cs.k / cs.s;
                for (int py = 0; py < cs.s; ++py)
                    for (int px = 0; px < cs.s; ++px)
                        push(w, c.w.conv_wd[i][py * cs.s + px], cs.cin, T * T * cs.cout, T * T * cs.cout, 3, bf, cs.cin, cs.cout, cs.k, py, px, T, cs.s);
            }
        }
    }
    for (int id = 0; part == 1 && id < LIN_COUNT; ++id) {
        if (id == LIN_BOXH0 || id == LIN_ZH0) continue;
        const LinSpec& l = c.PL.lin[id];
        if (!l.in || !l.out) continue;        // (the convolutional variant has no dense0 / dense1 in its encoder, no dense1 / out in its decoder)
        const int ldf = round_up(l.in, 8);
        int head0 = -1;
        if (id == LIN_BOXH1) head0 = LIN_BOXH0;
        if (id == LIN_ZH1) head0 = LIN_ZH0;
        const int rows_total = l.out + (head0 >= 0 ? c.PL.lin[head0].out : 0);
        const int ldt = round_up(rows_total, 8);
        push(c.params + l.w, c.w.lin_wf[id], l.out, l.in, ldf, 0, bf);
        if (need_dgrad) push(c.params + l.w, c.w.lin_wt[id], l.in, l.out, ldt, 1, bf);
        if (head0 >= 0) {
            const LinSpec& h = c.PL.lin[head0];
            push(c.params + h.w, const_cast<void*>(bptr(c.w.lin_wf[id], (size_t)l.out * ldf, c.d.dtype)), h.out, h.in, ldf, 0, bf);
            if (need_dgrad) push(c.params + h.w, const_cast<void*>(bptr(c.w.lin_wt[id], (size_t)l.out, c.d.dtype)), h.in, h.out, ldt, 1, bf);
        }
    }
    if (part == 1 && c.use_chain) {
        auto pack = [&](int cw, int lin_id, int KT, int ksplit, int kpad0, int n_off) {
            const LinSpec& l = c.PL.lin[lin_id];
            PrepEntry e;
            memset(&e, 0, sizeof(e));
            e.src = c.params + l.w; e.dst = c.w.chain_w[cw]; e.rows = l.out; e.cols = l.in; e.mode = 4; e.bf16 = 1;
            e.KT = KT; e.ksplit = ksplit; e.kpad0 = kpad0; e.n_off = n_off;
            es.push_back(e);
        };
        const int fc = c.L.F + c.L.CTX;   // 324
        auto pack_lo = [&](int cw, int lin_id, int KT, int ksplit, int kpad0, int n_off) {      // same pack, low parts (bf16 = 2)
            pack(cw, lin_id, KT, ksplit, kpad0, n_off);
            es.back().dst = c.w.chain_wlo[cw]; es.back().bf16 = 2;
        };
        pack_lo(CW_BOX0, LIN_BOX0, 11, fc, 352, 0);
        pack_lo(CW_BOX1, LIN_BOX1, 4, SP_H, 128, 0);
        pack_lo(CW_BOXH, LIN_BOXH1, 4, SP_H, 128, 0); pack_lo(CW_BOXH, LIN_BOXH0, 4, SP_H, 128, c.L.NP);
        pack(CW_BOX0, LIN_BOX0, 11, fc, 352, 0);
        pack(CW_BOX1, LIN_BOX1, 4, SP_H, 128, 0);
        pack(CW_BOXH, LIN_BOXH1, 4, SP_H, 128, 0); pack(CW_BOXH, LIN_BOXH0, 4, SP_H, 128, c.L.NP);
        pack(CW_ENC0, LIN_ENC0, 25, 784, 800, 0);
        pack(CW_ENC1, LIN_ENC1, 8, 256, 256, 0);
        pack(CW_ENC2, LIN_ENC2, 4, 128, 128, 0);
        pack(CW_Z0, LIN_Z0, 16, fc, 352, 0);
        pack(CW_Z1, LIN_Z1, 4, SP_H, 128, 0);
        pack(CW_ZH, LIN_ZH1, 4, SP_H, 128, 0); pack(CW_ZH, LIN_ZH0, 4, SP_H, 128, c.L.NP);
        pack(CW_OBJ0, LIN_OBJ0, 16, fc, 352, 0);
        pack(CW_OBJ1, LIN_OBJ1, 4, SP_H, 128, 0);
        pack(CW_OBJ2, LIN_OBJ2, 4, SP_H, 128, 0);
        if (need_dgrad) {
            auto packt = [&](int cw, int lin_id, int KT, int k_off) {
                const LinSpec& l = c.PL.lin[lin_id];
                PrepEntry e;
                memset(&e, 0, sizeof(e));
                e.src = c.params + l.w; e.dst = c.w.chain_wt[cw]; e.rows = l.out; e.cols = l.in; e.mode = 5; e.bf16 = 1;
                e.KT = KT; e.n_off = k_off;
                es.push_back(e);
            };
            packt(CW_BOX0, LIN_BOX0, 4, 0); packt(CW_BOX1, LIN_BOX1, 4, 0);
            packt(CW_BOXH, LIN_BOXH1, 4, 0); packt(CW_BOXH, LIN_BOXH0, 4, c.L.NP);
            packt(CW_ENC0, LIN_ENC0, 8, 0); packt(CW_ENC1, LIN_ENC1, 4, 0); packt(CW_ENC2, LIN_ENC2, 4, 0);
            packt(CW_Z0, LIN_Z0, 4, 0); packt(CW_Z1, LIN_Z1, 4, 0);
            packt(CW_ZH, LIN_ZH1, 4, 0); packt(CW_ZH, LIN_ZH0, 4, c.L.NP);
            packt(CW_OBJ0, LIN_OBJ0, 4, 0); packt(CW_OBJ1, LIN_OBJ1, 4, 0);
        }
    }
    if (part == 1 && c.use_dec_fused)
        TRY(dec_fused_pack(c.params + c.PL.lin[LIN_DEC0].w, c.params + c.PL.lin[LIN_DEC1].w, c.params + c.PL.lin[LIN_DEC2].w, c.d.A,
                           c.d.P * c.d.P * (c.d.C + 1), c.d.obj_logit_scale, c.d.alpha_logit_scale, c.w.dec_stream, c.s));
    if (part == 1) {
        push(c.params + c.PL.lin[LIN_BOXH1].b, c.w.bias_boxh, 1, c.L.NP, c.L.NP + 8, 0, 0);
        push(c.params + c.PL.lin[LIN_BOXH0].b, c.w.bias_boxh + c.L.NP, 1, 8, 8, 0, 0);
        push(c.params + c.PL.lin[LIN_ZH1].b, c.w.bias_zh, 1, c.L.NP, c.L.NP + 8, 0, 0);
        push(c.params + c.PL.lin[LIN_ZH0].b, c.w.bias_zh + c.L.NP, 1, 2, 8, 0, 0);
    }
    for (size_t i = 0; i < es.size(); i += SP_MAX_PREP) {
        PrepTable T;
        T.n = (int)std::min((size_t)SP_MAX_PREP, es.size() - i);
        for (int j = 0; j < T.n; ++j) T.e[j] = es[i + j];
        TRY(misc_prep(T, c.s));
    }
    return SPAIR_OK;
}

// ---- backbone ------------------------------------------------------------------------------------------
static ConvDesc fwd_desc(const ConvSpec& cs) {
    ConvDesc cd;
    cd.Hin = cs.hin; cd.Win = cs.hin; cd.Cin = cs.cin; cd.Hout = cs.hout; cd.Wout = cs.hout; cd.kh = cs.k; cd.kw = cs.k;
    cd.sy = cs.s; cd.sx = cs.s; cd.dky = 1; cd.dkx = 1; cd.oy = 0; cd.ox = 0;
    return cd;
}

// First layer of the trailing run of 1x1 convolutions that pointwise.hip can fuse (128 channels in, 128 out except the last,
// at most 4 layers), or n_conv if there is none.
static int pw_stack_first(const Ctx& c) {
    const int n = c.PL.n_conv;
    if (c.d.dtype != SPAIR_BF16) return n;
    int i0 = n;
    for (int i = n - 1; i >= 1; --i) {
        const ConvSpec& cs = c.PL.conv[i];
        const bool last = (i == n - 1);
        if (cs.k != 1 || cs.s != 1 || cs.cin != 128 || (last ? cs.cout > 128 : cs.cout != 128) || n - i > 4) break;
        if (c.PL.conv[i - 1].cout != 128) break;
        i0 = i;
    }
    return (n - i0 >= 2) ? i0 : n;
}

// input padding + the stem conv: they read the fp32 parameters directly, so they do not wait for the weight preparation
static int backbone_stem_fwd(Ctx& c) {
    const SpairDims& d = c.d;
    const int b16 = d.dtype == SPAIR_BF16;
    const ConvSpec& c0 = c.PL.conv[0];
    // the fast stem reads the unpadded image itself (spair_forward makes the padded copy on the helper stream, for the weight gradient)
    if (!misc_conv0_reads_unpadded(d.B, c0.hin, d.C, c0.k, c0.cout))
        TRY(misc_pad_input(c.x, c.w.xpad, d.B, d.C, d.I, d.pad_pre, d.I + d.pad_pre + d.pad_post, c.s));
    return misc_conv0_fwd(c.x, c.w.xpad, c.params + c0.w, c.params + c0.b, c.w.act[0], d.B, d.I, d.pad_pre, c0.hin, d.C, c0.k, c0.s, c0.hout,
                          c0.cout, b16, c.s, c.w.act0_bits);
}
static int backbone_fwd(Ctx& c) {
    const SpairDims& d = c.d;
    const int b16 = d.dtype == SPAIR_BF16;
    const int pw0 = pw_stack_first(c);
    for (int i = 1; i < c.PL.n_conv; ++i) {
        const ConvSpec& cs = c.PL.conv[i];
        const bool last = (i == c.PL.n_conv - 1);
        const int M = d.B * cs.hout * cs.hout, K = cs.k * cs.k * cs.cin;
        if (i == pw0) {   // the trailing 1x1 layers run as one fused per-pixel MLP
            const void* W[4]; const float* bias[4]; void* Y[4]; int ldw[4], cout[4];
            const int Lp = c.PL.n_conv - pw0;
            for (int l = 0; l < Lp; ++l) {
                const ConvSpec& q = c.PL.conv[pw0 + l];
                W[l] = c.w.conv_wf[pw0 + l]; ldw[l] = round_up(q.cin, 8); cout[l] = q.cout; bias[l] = c.params + q.b; Y[l] = c.w.act[pw0 + l];
            }
            TRY(spair_pw_stack_fwd16(c.w.act[pw0 - 1], W, ldw, cout, bias, Y, c.w.feat, c.w.ld_feat, M, Lp, c.s));
            break;
        }
        float* out = last ? c.w.feat : c.w.act[i];
        const int ldc = last ? c.w.ld_feat : cs.cout;
        const ConvDesc cd = fwd_desc(cs);
        if (b16 && !last && cs.k > 1 && conv_kperm(c, cs) && !(c.st.flags & 32)) {
            // 128 -> 128 channel 4x4 / stride-2 layers: the patch-resident kernel (conv_s2.hip), 2.3x fewer operand bytes from L2
            ProfScope ps(i == 1 ? PS_CONV1_FWD : -1, c.s);
            const int rc = conv_s2k4_patch_fwd16(c.w.act[i - 1], c.w.conv_wf[i], c.params + cs.b, out, d.B, cs.hin, cs.hout, cs.cin, cs.cout, cs.k, cs.s, c.s,
                                                 c.w.act_bits[i]);      // (also with train = 0: the backward picks its gate from the geometry alone)
            if (rc == SPAIR_OK) continue;
            if (rc != SPAIR_ERR_UNSUPPORTED) return rc;
        }
        if (b16) {   // activations stored as bf16; the feature map handed to the per-cell chain stays fp32
            ProfScope ps(i == 1 ? PS_CONV1_FWD : -1, c.s);
            TRY(nt16(c, c.w.act[i - 1], cs.cin, c.w.conv_wf[i], round_up(K, 8), out, ldc, last ? 0 : 1, M, cs.cout, round_up(K, 8),
                     c.params + cs.b, nullptr, 0, last ? 0 : 1, cs.k == 1 ? nullptr : &cd, nullptr, cs.k == 1 ? nullptr : &cs));
        } else if (cs.k == 1) {
            TRY(nt(c, c.w.act[i - 1], cs.cin, c.w.conv_wf[i], round_up(K, 8), out, ldc, M, cs.cout, round_up(K, 8), c.params + cs.b, nullptr, 0, last ? 0 : 1));
        } else {
            GemmNT g;
            memset(&g, 0, sizeof(g));
            g.A = c.w.act[i - 1]; g.B = c.w.conv_wf[i]; g.ldb = round_up(K, 8); g.C = out; g.ldc = ldc; g.M = M; g.N = cs.cout; g.K = K;
            g.bias = c.params + cs.b; g.relu = last ? 0 : 1; g.conv = cd;
            ProfScope ps(i == 1 ? PS_CONV1_FWD : -1, c.s);
            TRY(spair_gemm_nt_impl(g, true, d.dtype, c.s));
        }
    }
    return SPAIR_OK;
}

static int backbone_bwd16(Ctx& c, float* grads) {
    const SpairDims& d = c.d;
    const int last = c.PL.n_conv - 1;
    bool stem_fused = false;
    const int N = d.B * d.G * d.G;
    if (!c.use_chain) TRY(spair_to_bf16(c.w.dfeat, c.w.ld_feat, c.w.dfeat16, c.w.ld_feat, N, c.w.ld_feat, c.s));   // the fused chain writes bf16 itself
    const int pw0 = pw_stack_first(c);
    if (pw0 <= last) {   // data gradients of the trailing 1x1 layers: one fused kernel, top layer first
        const void* Wd[4]; const void* gate[4]; void* dX[4]; int ldw[4], cout[4];
        const int Lp = last - pw0 + 1;
        for (int l = 0; l < Lp; ++l) {
            const int i = last - l;
            const ConvSpec& q = c.PL.conv[i];
            Wd[l] = c.w.conv_wd[i][0]; ldw[l] = round_up(q.cout, 8); cout[l] = q.cout; gate[l] = c.w.act[i - 1]; dX[l] = c.w.dact[i - 1];
        }
        TRY(spair_pw_stack_bwd16(c.w.dfeat16, c.w.ld_feat, c.PL.conv[last].cout, Wd, ldw, cout, gate, dX, N, Lp, c.s));
    }
    if (pw0 <= last) {   // ... and their weight / bias gradients as ONE grouped split-K GEMM + one reduce pass
        GemmTN g;
        memset(&g, 0, sizeof(g));
        g.ngroup = last - pw0 + 1;
        for (int q = 0; q < g.ngroup; ++q) {
            const int i = pw0 + q;
            const ConvSpec& cs = c.PL.conv[i];
            GemmTN::Tile& t = g.tile[q];
            t.A = (i == last) ? c.w.dfeat16 : (const void*)c.w.dact[i];
            t.lda = (i == last) ? c.w.ld_feat : cs.cout;
            t.B = c.w.act[i - 1]; t.ldb = cs.cin;
            t.C = grads + cs.w; t.ldc = cs.cin; t.colsum = grads + cs.b;
            t.M = round_up(cs.cout, 8); t.N = cs.cin; t.Mstore = cs.cout; t.Nstore = cs.cin;
        }
        g.R = N;
        g.part = c.tn_scratch ? c.tn_scratch : c.w.tn_part; g.part_cap = SPAIR_TN_PART_FLOATS;
        TRY(spair_gemm_tn16_impl(g, false, true, c.s));
    }
    for (int i = last; i >= 1; --i) {
        const ConvSpec& cs = c.PL.conv[i];
        const int M = d.B * cs.hout * cs.hout;
        const void* dout = (i == last) ? c.w.dfeat16 : (const void*)c.w.dact[i];
        const int ldd = (i == last) ? c.w.ld_feat : cs.cout;
        const void* in = c.w.act[i - 1];
        const ConvDesc cd = fwd_desc(cs);
        if (cs.k == 1) {
            if (i >= pw0) continue;
            TRY(tn16(c, dout, ldd, cs.cout, in, cs.cin, cs.cin, true, grads + cs.w, cs.cin, M, grads + cs.b));
            const int Kd = round_up(cs.cout, 8);
            if (i < pw0) TRY(nt16(c, dout, ldd, c.w.conv_wd[i][0], Kd, c.w.dact[i - 1], cs.cin, 1, M, cs.cin, Kd, nullptr, in, cs.cin, 0));
        } else {
            const int K = cs.k * cs.k * cs.cin;
            TRY(tn16(c, dout, ldd, cs.cout, in, 0, K, true, grads + cs.w, K, M, grads + cs.b, &cd, cs.cin, cs.k * cs.k));
            const int T = cs.k / cs.s;
            if (cs.k == 4 && cs.s == 2 && cs.cin == 128 && cs.cout == 128 && cs.hin == 2 * (cs.hout + 1) && !(c.st.flags & 32)) {
                // patch-resident data gradient (conv_s2_dgrad.hip): the d-out neighbourhood of a tile is staged once for all 4 parity classes x 4 taps
                const ConvSpec& c0 = c.PL.conv[0];
                const bool want_stem = i == 1 && c0.cin == 1 && c0.k == 4 && c0.cout == 128 && c0.hout == cs.hin && !(c.st.flags & 8);
                const void* wd4[4] = {c.w.conv_wd[i][0], c.w.conv_wd[i][1], c.w.conv_wd[i][2], c.w.conv_wd[i][3]};
                // conv_1's gate as the stem kernel's sign bits (20 MB instead of the 321-MB activation) whenever that kernel wrote them
                const void* gbits = nullptr;
                if (i == 1) { if (c0.hout == cs.hin) gbits = c.w.act0_bits; }
                else {          // the layer below ran on the patch-resident forward kernel (same test as backbone_fwd): it left its mask
                    const ConvSpec& lo = c.PL.conv[i - 1];
                    if (lo.k > 1 && conv_kperm(c, lo) && conv_s2k4_patch_fwd16_fits(d.B, lo.hin, lo.hout, lo.cin, lo.cout, lo.k, lo.s)) gbits = c.w.act_bits[i - 1];
                }
                int rc = conv_s2k4_patch_dgrad16(dout, wd4, in, c.w.dact[i - 1], d.B, cs.hout, cs.hin, cs.cin, cs.cout, cs.k, cs.s,
                                                 want_stem ? c.w.xpad : nullptr, c0.hin, c0.s, want_stem ? c.w.tn_part : nullptr,
                                                 SPAIR_TN_PART_FLOATS, grads + c0.w, grads + c0.b, c.s, gbits);
                if (rc == SPAIR_OK) { if (want_stem) stem_fused = true; continue; }
                if (rc == SPAIR_ERR_UNSUPPORTED && want_stem) {      // the stem fusion alone was refused: same kernel, d act0 to HBM, stem wgrad below
                    rc = conv_s2k4_patch_dgrad16(dout, wd4, in, c.w.dact[i - 1], d.B, cs.hout, cs.hin, cs.cin, cs.cout, cs.k, cs.s, nullptr, 0, 0,
                                                 nullptr, 0, nullptr, nullptr, c.s, gbits);
                    if (rc == SPAIR_OK) continue;
                }
                if (rc != SPAIR_ERR_UNSUPPORTED) return rc;
            }
            if (cs.hin % cs.s == 0 && cs.s * cs.s <= 4) {
                // all output-parity classes have the same size: one launch, blockIdx.z = class (4 launches of 1.1 rounds of
                // resident blocks each ran as 2 rounds: conv2's data-gradient took 0.37 ms for 34 GFLOP)
                const int Hc = cs.hin / cs.s;
                GemmNT g;
                memset(&g, 0, sizeof(g));
                g.A = reinterpret_cast<const float*>(dout); g.ldb = T * T * cs.cout; g.C = c.w.dact[i - 1]; g.ldc = cs.cin; g.c_bf16 = 1;
                g.M = d.B * Hc * Hc; g.N = cs.cin; g.K = T * T * cs.cout;
                g.mask = reinterpret_cast<const float*>(in); g.ldmask = cs.cin; g.mask_bf16 = 1;
                g.conv.Hin = cs.hout; g.conv.Win = cs.hout; g.conv.Cin = cs.cout; g.conv.Hout = Hc; g.conv.Wout = Hc; g.conv.kh = T; g.conv.kw = T;
                g.conv.sy = 1; g.conv.sx = 1; g.conv.dky = -1; g.conv.dkx = -1; g.conv.oy = 0; g.conv.ox = 0;
                g.use_cmap = 1;
                g.cmap.Hout = Hc; g.cmap.Wout = Hc; g.cmap.Hc = cs.hin; g.cmap.Wc = cs.hin; g.cmap.osy = cs.s; g.cmap.osx = cs.s;
                g.nz = cs.s * cs.s;
                for (int q = 0; q < g.nz; ++q) g.Bz[q] = c.w.conv_wd[i][q];
                g.B = g.Bz[0];
                if (i == 1) {     // conv_1: take the stem's weight gradient from the tile in LDS; d act0 never reaches HBM
                    const ConvSpec& c0 = c.PL.conv[0];
                    g.stem_xp = c.w.xpad; g.stem_hin = c0.hin; g.stem_s = c0.s; g.stem_dw = grads + c0.w; g.stem_db = grads + c0.b;
                    g.stem_part = c.w.tn_part; g.stem_part_cap = SPAIR_TN_PART_FLOATS;
                    stem_fused = c0.cin == 1 && c0.k == 4 && c0.cout == 128 && c0.hout == cs.hin && !(c.st.flags & 8) &&
                                 spair_nt16_stem_fusable(g, g.stem_part_cap);
                    if (!stem_fused) g.stem_part = nullptr;
                }
                TRY(spair_gemm_nt16_impl(g, true, c.s));
            } else
            for (int py = 0; py < cs.s; ++py)
                for (int px = 0; px < cs.s; ++px) {
                    const int Hc = (cs.hin - py + cs.s - 1) / cs.s, Wc = (cs.hin - px + cs.s - 1) / cs.s;
                    if (Hc <= 0 || Wc <= 0) continue;
                    ConvDesc dd;
                    dd.Hin = cs.hout; dd.Win = cs.hout; dd.Cin = cs.cout; dd.Hout = Hc; dd.Wout = Wc; dd.kh = T; dd.kw = T;
                    dd.sy = 1; dd.sx = 1; dd.dky = -1; dd.dkx = -1; dd.oy = 0; dd.ox = 0;
                    RowMap rm;
                    rm.Hout = Hc; rm.Wout = Wc; rm.Hc = cs.hin; rm.Wc = cs.hin; rm.osy = cs.s; rm.osx = cs.s; rm.ooy = py; rm.oox = px;
                    TRY(nt16(c, dout, 0, c.w.conv_wd[i][py * cs.s + px], T * T * cs.cout, c.w.dact[i - 1], cs.cin, 1, d.B * Hc * Wc, cs.cin,
                             T * T * cs.cout, nullptr, in, cs.cin, 0, &dd, &rm));
                }
        }
    }
    if (!stem_fused) {   // first layer: A = d act0 (bf16), B gathered element-wise from the padded fp32 input
        const ConvSpec& c0 = c.PL.conv[0];
        const ConvDesc cd = fwd_desc(c0);
        const int K = c0.k * c0.k * c0.cin;
        if (c0.cin == 1 && c0.k == 4 && c0.cout == 128) {
            TRY(spair_stem_wgrad16_impl(c.w.dact[0], c.w.xpad, grads + c0.w, grads + c0.b, c.w.tn_part, SPAIR_TN_PART_FLOATS, d.B, c0.hin,
                                        c0.s, c0.hout, c.s));
        } else {
            TRY(tn16(c, c.w.dact[0], c0.cout, c0.cout, c.w.xpad, 0, K, false, grads + c0.w, K, d.B * c0.hout * c0.hout, grads + c0.b, &cd,
                     c0.cin, c0.k * c0.k));
        }
    }
    return SPAIR_OK;
}

static int backbone_bwd(Ctx& c, float* grads) {
    const SpairDims& d = c.d;
    if (d.dtype == SPAIR_BF16) return backbone_bwd16(c, grads);
    const int last = c.PL.n_conv - 1;
    for (int i = last; i >= 1; --i) {
        const ConvSpec& cs = c.PL.conv[i];
        const int M = d.B * cs.hout * cs.hout;
        const float* dout = (i == last) ? c.w.dfeat : c.w.dact[i];
        const int ldd = (i == last) ? c.w.ld_feat : cs.cout;
        const float* in = c.w.act[i - 1];
        // weight + bias gradients
        if (cs.k == 1) {
            TRY(tn(c, dout, ldd, cs.cout, in, cs.cin, cs.cin, grads + cs.w, cs.cin, M, grads + cs.b));
        } else {
            GemmTN g;
            memset(&g, 0, sizeof(g));
            const int K = cs.k * cs.k * cs.cin;
            g.A = dout; g.lda = ldd; g.B = in; g.C = grads + cs.w; g.ldc = K; g.M = round_up(cs.cout, 4); g.N = K; g.Mstore = cs.cout; g.Nstore = K;
            g.R = M; g.cw_cin = cs.cin; g.cw_taps = cs.k * cs.k; g.conv = fwd_desc(cs); g.colsum_out = grads + cs.b;
            TRY(spair_gemm_tn_impl(g, true, c.d.dtype, c.s));
        }
        // data gradient into dact[i-1] (masked by relu of act[i-1])
        if (cs.k == 1) {
            const int Kd = round_up(cs.cout, 8);
            TRY(nt(c, dout, ldd, c.w.conv_wd[i][0], Kd, c.w.dact[i - 1], cs.cin, M, cs.cin, Kd, nullptr, in, cs.cin, 0));
        } else {
            const int T = cs.k / cs.s;
            for (int py = 0; py < cs.s; ++py)
                for (int px = 0; px < cs.s; ++px) {
                    const int Hc = (cs.hin - py + cs.s - 1) / cs.s, Wc = (cs.hin - px + cs.s - 1) / cs.s;
                    if (Hc <= 0 || Wc <= 0) continue;
                    GemmNT g;
                    memset(&g, 0, sizeof(g));
                    g.A = dout; g.B = c.w.conv_wd[i][py * cs.s + px]; g.ldb = T * T * cs.cout; g.C = c.w.dact[i - 1]; g.ldc = cs.cin;
                    g.M = d.B * Hc * Wc; g.N = cs.cin; g.K = T * T * cs.cout; g.mask = in; g.ldmask = cs.cin;
                    g.conv.Hin = cs.hout; g.conv.Win = cs.hout; g.conv.Cin = cs.cout; g.conv.Hout = Hc; g.conv.Wout = Wc;
                    g.conv.kh = T; g.conv.kw = T; g.conv.sy = 1; g.conv.sx = 1; g.conv.dky = -1; g.conv.dkx = -1; g.conv.oy = 0; g.conv.ox = 0;
                    g.use_cmap = 1;
                    g.cmap.Hout = Hc; g.cmap.Wout = Wc; g.cmap.Hc = cs.hin; g.cmap.Wc = cs.hin; g.cmap.osy = cs.s; g.cmap.osx = cs.s;
                    g.cmap.ooy = py; g.cmap.oox = px;
                    TRY(spair_gemm_nt_impl(g, true, d.dtype, c.s));
                }
        }
    }
    {   // first layer (Cin = image channels): the same TN GEMM, B gathered element-wise from the padded input
        const ConvSpec& c0 = c.PL.conv[0];
        GemmTN g;
        memset(&g, 0, sizeof(g));
        const int K = c0.k * c0.k * c0.cin;
        g.A = c.w.dact[0]; g.lda = c0.cout; g.B = c.w.xpad; g.C = grads + c0.w; g.ldc = K; g.M = round_up(c0.cout, 4); g.N = round_up(K, 4);
        g.Mstore = c0.cout; g.Nstore = K; g.R = d.B * c0.hout * c0.hout; g.cw_cin = c0.cin; g.cw_taps = c0.k * c0.k; g.conv = fwd_desc(c0);
        g.colsum_out = grads + c0.b;
        TRY(spair_gemm_tn_impl(g, true, c.d.dtype, c.s));
    }
    return SPAIR_OK;
}

// ---- forward ---------------------------------------------------------------------------------------------
static int fwd_lin(Ctx& c, int id, const float* A, int lda, float* C, int ldc, int r0, int R, const float* bias, int N, int relu) {
    const int K = round_up(c.PL.lin[id].in, 8);
    return nt(c, A + (size_t)r0 * lda, lda, c.w.lin_wf[id], K, C + (size_t)r0 * ldc, ldc, R, N, K, bias, nullptr, 0, relu);
}

// The box network's forward in the bf16 step runs at (nearly) fp32 precision -- split-bf16 products in the fused kernel (chain.hip), the raw
// fp32 parameters here in the per-wavefront reference path: its outputs place the glimpse and the sprite on the pixel grid, and bf16 operands
// there move the reconstruction by up to 0.05 and the box / encoder gradients' direction to cos 0.92 against the reference
// (tools/exp/f32nets_table.py; models.py:76-79,322-381).
static int fwd_lin_f32(Ctx& c, int id, const float* A, int lda, float* C, int ldc, int c0, int r0, int R, int relu) {
    const LinSpec& l = c.PL.lin[id];
    GemmNT g;
    memset(&g, 0, sizeof(g));
    g.A = A + (size_t)r0 * lda; g.lda = lda; g.B = c.params + l.w; g.ldb = l.in; g.C = C + (size_t)r0 * ldc + c0; g.ldc = ldc;
    g.M = R; g.N = l.out; g.K = l.in; g.bias = c.params + l.b; g.relu = relu;
    return spair_gemm_nt_impl(g, false, SPAIR_F32, c.s);
}

// ---- convolutional object encoder / decoder variant (SpairDims.obj_conv; objconv.hip) --------------------------------------------------
// Parity unpinned: the reference's ObjectConvEncoder / ObjectConvDecoder (models.py:606-665) cannot run; the layer sizes follow
// CONV_OBJECT_ENCODER_TOPOLOGY's own comments (config.py:15-20).  Activations are per-object NHWC tensors, except the encoder's last
// layer and the decoder's Linear output, which are stored in (C,H,W) order -- the order nn.Flatten / .view give the Linear layers.
static OcTensor oc_enc_act(const Ctx& c, int i, bool grad) {
    const ConvSpec& e = c.PL.oc_enc[i];
    float* p = grad ? c.w.oc_dea[i] : c.w.oc_ea[i];
    const long long rs = (long long)e.hout * e.hout * e.cout;
    return i + 1 == c.PL.oc_n ? oc_chw(p, rs, e.hout, e.cout) : oc_hwc(p, rs, e.hout, e.cout);
}
static OcTensor oc_dec_act(const Ctx& c, int i, bool grad) {      // input of transposed conv i + 1 (i = -1: the Linear's output)
    if (i < 0) return oc_chw(grad ? c.w.oc_ddh : c.w.oc_dh, c.PL.oc_flat, c.PL.oc_dec[0].hin, c.PL.oc_dec[0].cin);
    const ConvSpec& t = c.PL.oc_dec[i];
    return oc_hwc(grad ? c.w.oc_dda[i] : c.w.oc_da[i], (long long)t.hout * t.hout * t.cout, t.hout, t.cout);
}
static int bwd_lin(Ctx& c, int id, int out_total, const float* dOut, int ldo, float* dX, int ldx, int r0, int R, const float* mask, int ldmask);
static int wgrad_lin(Ctx& c, int id, const float* dOut, int ldo, const float* In, int ldi, float* grads, int R);
// Weight gradient of one layer: G[cs][cb][ky][kx] += sum small * big-through-the-taps.  When both operands are dense NHWC tensors it is the
// split-K TN product of gemm.hip with the big tensor gathered as the implicit-GEMM operand (the kernel of the backbone's weight gradients:
// conv0's 1.6 M positions took 1.1 ms as objconv's walk); the (C,H,W)-ordered layers at the 2 x 2 end keep the walk.
static int oc_layer_wgrad(Ctx& c, const OcTensor& sm, const OcTensor& bg, float* G, float* bias_small, int k, int s, long long R) {
    auto dense = [](const OcTensor& t) { return t.cs == 1 && t.xs == t.C && t.ys == t.H * t.C && t.rs == (long long)t.H * t.H * t.C; };
    const long long rows = R * sm.H * sm.H;
    if (dense(sm) && dense(bg) && (sm.C & 3) == 0 && rows < 0x7fffffffll && R * bg.H * bg.H * bg.C < 0x7fffffffll) {
        GemmTN g;
        memset(&g, 0, sizeof(g));
        const int K = k * k * bg.C;
        g.A = sm.p; g.lda = sm.C; g.B = bg.p; g.C = G; g.ldc = K; g.M = sm.C; g.N = round_up(K, 4); g.Mstore = sm.C; g.Nstore = K;
        g.R = (int)rows; g.cw_cin = bg.C; g.cw_taps = k * k; g.colsum_out = bias_small;
        g.conv.Hin = bg.H; g.conv.Win = bg.H; g.conv.Cin = bg.C; g.conv.Hout = sm.H; g.conv.Wout = sm.H; g.conv.kh = k; g.conv.kw = k;
        g.conv.sy = s; g.conv.sx = s; g.conv.dky = 1; g.conv.dkx = 1; g.conv.oy = 0; g.conv.ox = 0;
        return spair_gemm_tn_impl(g, true, SPAIR_F32, c.s);
    }
    return oc_wgrad(sm, bg, G, bias_small, k, s, R, c.s);
}
static int oc_encoder_fwd(Ctx& c, int r0, int R) {
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const int n = c.PL.oc_n;
    OcTensor in = oc_hwc(P.glimpse, L.ld_gl, c.d.P, c.d.C);
    for (int i = 0; i < n; ++i) {
        const ConvSpec& e = c.PL.oc_enc[i];
        const OcTensor out = oc_enc_act(c, i, false);
        TRY(oc_gather(false, oc_rows(in, r0), c.params + e.w, c.params + e.b, oc_rows(out, r0), OcTensor{}, e.k, e.s, 1, R, c.s));
        in = out;
    }
    return fwd_lin(c, LIN_ENC2, c.w.oc_ea[n - 1], c.PL.oc_flat, P.Oe, L.ld_oe, r0, R, c.params + c.PL.lin[LIN_ENC2].b, 2 * L.A, 0);
}
static int oc_encoder_bwd(Ctx& c, int r0, int R) {
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const int n = c.PL.oc_n, flat = c.PL.oc_flat;
    TRY(bwd_lin(c, LIN_ENC2, 2 * L.A, P.dOe, L.ld_oe, c.w.oc_dea[n - 1], flat, r0, R, c.w.oc_ea[n - 1], flat));
    for (int i = n - 1; i >= 0; --i) {
        const ConvSpec& e = c.PL.oc_enc[i];
        const OcTensor din = i ? oc_enc_act(c, i - 1, true) : oc_hwc(P.dGl, L.ld_gl, c.d.P, c.d.C);
        const OcTensor gate = i ? oc_enc_act(c, i - 1, false) : OcTensor{};
        TRY(oc_gather(true, oc_rows(oc_enc_act(c, i, true), r0), c.params + e.w, nullptr, oc_rows(din, r0), oc_rows(gate, r0), e.k, e.s, 0, R, c.s));
    }
    return SPAIR_OK;
}
static int oc_encoder_wgrad(Ctx& c, float* grads) {
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const int n = c.PL.oc_n;
    for (int i = 0; i < n; ++i) {
        const ConvSpec& e = c.PL.oc_enc[i];
        const OcTensor big = i ? oc_enc_act(c, i - 1, false) : oc_hwc(P.glimpse, L.ld_gl, c.d.P, c.d.C);
        TRY(oc_layer_wgrad(c, oc_enc_act(c, i, true), big, grads + e.w, grads + e.b, e.k, e.s, L.N));
    }
    return wgrad_lin(c, LIN_ENC2, P.dOe, L.ld_oe, c.w.oc_ea[n - 1], c.PL.oc_flat, grads, L.N);
}
static int oc_decoder_fwd(Ctx& c) {
    const CellLayout& L = c.L;
    const int n = c.PL.oc_n, flat = c.PL.oc_flat, per = c.d.P * c.d.P * (c.d.C + 1);
    TRY(fwd_lin(c, LIN_DEC0, c.w.Za, L.ld_rec, c.w.oc_dh, flat, 0, L.N, c.params + c.PL.lin[LIN_DEC0].b, flat, 0));
    for (int i = 0; i < n; ++i) {
        const ConvSpec& t = c.PL.oc_dec[i];
        const OcTensor out = i + 1 < n ? oc_dec_act(c, i, false) : oc_hwc(c.w.S, c.w.ld_s, t.hout, t.cout);
        TRY(oc_gather(true, oc_dec_act(c, i - 1, false), c.params + t.w, c.params + t.b, out, OcTensor{}, t.k, t.s, i + 1 < n, L.N, c.s));
    }
    return render_sprite_act(c.w.S, c.w.ld_s, L.N, per, c.d.C + 1, c.d.obj_logit_scale, c.d.alpha_logit_scale, c.d.alpha_logit_bias, c.s);
}
static int oc_decoder_bwd(Ctx& c, float* grads) {
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const int n = c.PL.oc_n, flat = c.PL.oc_flat;
    for (int i = n - 1; i >= 0; --i) {
        const ConvSpec& t = c.PL.oc_dec[i];
        const OcTensor dout = i + 1 < n ? oc_dec_act(c, i, true) : oc_hwc(c.w.dLog, c.w.ld_s, t.hout, t.cout);
        const OcTensor in = oc_dec_act(c, i - 1, false);
        TRY(oc_layer_wgrad(c, in, dout, grads + t.w, nullptr, t.k, t.s, L.N));
        TRY(spair_colsum_impl(dout.p, t.cout, L.N * t.hout * t.hout, t.cout, grads + t.b, c.s));
        TRY(oc_gather(false, dout, c.params + t.w, nullptr, oc_dec_act(c, i - 1, true), i ? in : OcTensor{}, t.k, t.s, 0, L.N, c.s));
    }
    TRY(wgrad_lin(c, LIN_DEC0, c.w.oc_ddh, flat, c.w.Za, L.ld_rec, grads, L.N));
    return bwd_lin(c, LIN_DEC0, flat, c.w.oc_ddh, flat, P.g_attr_r, L.ld_rec, 0, L.N, nullptr, 0);
}

static int cells_fwd(Ctx& c) {
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const ParamLayout& PL = c.PL;
    const float* pr = c.params;
    if (c.use_chain) {
        ChainArgs a;
        a.L = L; a.P = P; a.H = c.H;
        for (int i = 0; i < CW_COUNT; ++i) a.w[i] = reinterpret_cast<const uint4*>(c.w.chain_w[i]);
        for (int i = 0; i < 3; ++i) a.wlo[i] = reinterpret_cast<const uint4*>(c.w.chain_wlo[i]);
        a.bias[CW_BOX0] = pr + PL.lin[LIN_BOX0].b; a.bias[CW_BOX1] = pr + PL.lin[LIN_BOX1].b; a.bias[CW_BOXH] = c.w.bias_boxh;
        a.bias[CW_ENC0] = pr + PL.lin[LIN_ENC0].b; a.bias[CW_ENC1] = pr + PL.lin[LIN_ENC1].b; a.bias[CW_ENC2] = pr + PL.lin[LIN_ENC2].b;
        a.bias[CW_Z0] = pr + PL.lin[LIN_Z0].b; a.bias[CW_Z1] = pr + PL.lin[LIN_Z1].b; a.bias[CW_ZH] = c.w.bias_zh;
        a.bias[CW_OBJ0] = pr + PL.lin[LIN_OBJ0].b; a.bias[CW_OBJ1] = pr + PL.lin[LIN_OBJ1].b; a.bias[CW_OBJ2] = pr + PL.lin[LIN_OBJ2].b;
        a.x = c.x; a.I = c.d.I; a.Pp = c.d.P; a.ac = c.d.align_corners;
        a.w_obj2 = pr + PL.lin[LIN_OBJ2].w; a.gedge = nullptr; a.stamps = (c.st.flags & 2) ? c.w.stamps : nullptr;
        a.nbands = chain_bands(c.d); a.sync = c.w.chain_sync; a.bnd_rec = c.w.bnd_rec; a.bnd_grad = c.w.bnd_grad;
        for (int i = 0; i < CW_COUNT; ++i) a.wt[i] = nullptr;
        return chain_fwd(a, c.s);
    }
    for (int t = 0; t < c.T; ++t) {
        const int r0 = c.dstart[t] * L.B, R = (c.dstart[t + 1] - c.dstart[t]) * L.B;
        TRY(cells_ctx_gather(L, P, r0, R, c.s));
        // z_where
        if (c.d.dtype == SPAIR_BF16 && (PL.lin[LIN_BOX0].in & 3) == 0) {
            TRY(fwd_lin_f32(c, LIN_BOX0, P.Xb, L.ld_xb, P.Hb1, SP_LDH, 0, r0, R, 1));
            TRY(fwd_lin_f32(c, LIN_BOX1, P.Hb1, SP_LDH, P.Hb2, SP_LDH, 0, r0, R, 1));
            TRY(fwd_lin_f32(c, LIN_BOXH1, P.Hb2, SP_LDH, P.Ob, L.ld_ob, 0, r0, R, 0));
            TRY(fwd_lin_f32(c, LIN_BOXH0, P.Hb2, SP_LDH, P.Ob, L.ld_ob, L.ob_lat, r0, R, 0));
        } else {
        TRY(fwd_lin(c, LIN_BOX0, P.Xb, L.ld_xb, P.Hb1, SP_LDH, r0, R, pr + PL.lin[LIN_BOX0].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_BOX1, P.Hb1, SP_LDH, P.Hb2, SP_LDH, r0, R, pr + PL.lin[LIN_BOX1].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_BOXH1, P.Hb2, SP_LDH, P.Ob, L.ld_ob, r0, R, c.w.bias_boxh, L.NP + 8, 0));
        }
        TRY(cells_box_sample(L, P, c.H, r0, R, c.s));
        // z_what
        { ProfScope ps(PS_STN_FWD, c.s); TRY(stn_glimpse_fwd(c.x, P.nbox, L.B, P.glimpse, L.ld_gl, r0, R, c.d.C, c.d.I, c.d.P, c.d.align_corners, chain_image_fp16(c.d), c.s)); }
        if (PL.oc_n) TRY(oc_encoder_fwd(c, r0, R));
        else {
        TRY(fwd_lin(c, LIN_ENC0, P.glimpse, L.ld_gl, P.He1, SP_ENC_H1, r0, R, pr + PL.lin[LIN_ENC0].b, SP_ENC_H1, 1));
        TRY(fwd_lin(c, LIN_ENC1, P.He1, SP_ENC_H1, P.He2, SP_ENC_H2, r0, R, pr + PL.lin[LIN_ENC1].b, SP_ENC_H2, 1));
        TRY(fwd_lin(c, LIN_ENC2, P.He2, SP_ENC_H2, P.Oe, L.ld_oe, r0, R, pr + PL.lin[LIN_ENC2].b, 2 * L.A, 0));
        }
        TRY(cells_attr_sample(L, P, r0, R, c.s));
        // z_depth
        TRY(fwd_lin(c, LIN_Z0, P.Xz, L.ld_x, P.Hz1, SP_LDH, r0, R, pr + PL.lin[LIN_Z0].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_Z1, P.Hz1, SP_LDH, P.Hz2, SP_LDH, r0, R, pr + PL.lin[LIN_Z1].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_ZH1, P.Hz2, SP_LDH, P.Oz, L.ld_oz, r0, R, c.w.bias_zh, L.NP + 2, 0));
        TRY(cells_depth_sample(L, P, c.H, r0, R, c.s));
        // z_pres
        TRY(fwd_lin(c, LIN_OBJ0, P.Xo, L.ld_x, P.Ho1, SP_LDH, r0, R, pr + PL.lin[LIN_OBJ0].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_OBJ1, P.Ho1, SP_LDH, P.Ho2, SP_LDH, r0, R, pr + PL.lin[LIN_OBJ1].b, SP_H, 1));
        TRY(fwd_lin(c, LIN_OBJ2, P.Ho2, SP_LDH, P.Oo, L.ld_oo, r0, R, pr + PL.lin[LIN_OBJ2].b, 1, 0));
        TRY(cells_pres_sample(L, P, c.H, r0, R, c.s));
    }
    return SPAIR_OK;
}

extern "C" int spair_forward(const SpairDims* d, const SpairStep* st, const float* params, const float* x, const float* eps_box,
                             const float* eps_attr, const float* eps_depth, const float* u_pres, void* workspace, float* loss_out,
                             float* recon, float* z_where, float* z_pres, void* stream) {
    Ctx c;
    TRY(make_ctx(c, d, st, params, x, eps_box, eps_attr, eps_depth, u_pres, workspace, stream));
    if (!loss_out || !recon || !z_where || !z_pres || !eps_box || !eps_attr || !eps_depth || !u_pres) return SPAIR_ERR_SHAPE;
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    P.z_where = z_where; P.z_pres = z_pres;
    SideStream* side = nullptr;
    if (!(st->flags & 4)) TRY(side_stream(side));
    std::unique_lock<std::mutex> enq_lock;
    if (side) enq_lock = std::unique_lock<std::mutex>(side->enq_mu);
    {   // tables + per-step weight copies (helper stream) beside the input padding and the stem conv (caller's stream)
        hipStream_t const main_s = c.s;
        if (side) { TRY(stream_link(main_s, side->s, side->ev[2])); c.s = side->s; }
        {
            ProfScope ps(PS_PREP, c.s);
            TRY(prep_weights(c, st->train != 0, 0));       // conv weights first: conv_1 waits for these only
            if (side && hipEventRecord(side->ev[3], side->s) != hipSuccess) return SPAIR_ERR_LAUNCH;
            TRY(cells_init_tables(d->G, c.L.LB, c.w.cell_h, c.w.cell_w, c.w.cidx, c.w.nbr, c.w.cons, c.w.diag_start, c.s));
            if (st->draw_noise)
                TRY(spair_noise_fill(d, st->noise_seed, const_cast<float*>(eps_box), const_cast<float*>(eps_attr), const_cast<float*>(eps_depth),
                                     const_cast<float*>(u_pres), c.s));
            TRY(prep_weights(c, st->train != 0, 1));
        }
        if (side && hipEventRecord(side->ev[4], side->s) != hipSuccess) return SPAIR_ERR_LAUNCH;
        // the padded copy is only read by the stem's weight gradient: it stays behind the event conv_1 waits on (the helper stream's
        // later joins order it before the backward)
        if (misc_conv0_reads_unpadded(d->B, c.PL.conv[0].hin, d->C, c.PL.conv[0].k, c.PL.conv[0].cout))
            TRY(misc_pad_input(x, c.w.xpad, d->B, d->C, d->I, d->pad_pre, d->I + d->pad_pre + d->pad_post, c.s));
        c.s = main_s;
    }
    const int ps_bb = prof_begin(PS_BACKBONE_FWD, c.s);
    TRY(backbone_stem_fwd(c));
    if (side && hipStreamWaitEvent(c.s, side->ev[3], 0) != hipSuccess) return SPAIR_ERR_LAUNCH;
    TRY(backbone_fwd(c));
    prof_end(ps_bb, c.s);
    if (side && hipStreamWaitEvent(c.s, side->ev[4], 0) != hipSuccess) return SPAIR_ERR_LAUNCH;      // tables, per-cell and decoder weights
    { ProfScope ps(PS_CELLS_FWD, c.s); TRY(cells_fwd(c)); }
    // the KL terms only need the cell chain's outputs: they run on the helper stream beside the decoder and the renderer.  (The count-prior
    // KL is one dependent chain per sample, ~0.17 ms beside the decoder at configs[1] and as long as decoder + renderer together: nothing
    // may sit in front of it on the helper stream -- the renderer's record kernel, 6 us, runs on the caller's stream instead.)
    int rc_prep = SPAIR_ERR_UNSUPPORTED;
    {
        hipStream_t ks = side ? side->s : c.s;
        if (side) TRY(stream_link(c.s, ks, side->ev[0]));
        { ProfScope ps(PS_COUNT_KL, ks); TRY(loss_count_kl(L, P, st->count_prior_prob, c.w.klp, ks)); }
        // the Gaussian KL sums: behind the count KL where that one hides beside the decoder and the renderer (grids up to 16 x 16); on wider
        // grids the count KL is the longer branch (0.34 against 0.23 ms at 32 x 32, B = 64) and the 18-us kernel goes to the caller's stream
        const bool gauss_on_main = side && d->G > 16;
        if (!gauss_on_main) TRY(loss_gauss_kl(L, P, c.H, c.w.kl_partial, ks));
        if (side && hipEventRecord(side->ev[1], ks) != hipSuccess) return SPAIR_ERR_LAUNCH;
        if (gauss_on_main) TRY(loss_gauss_kl(L, P, c.H, c.w.kl_partial, c.s));
    }
    if (d->dtype == SPAIR_BF16 && d->C == 1 && !d->obj_conv) {      // (the conv decoder's sprites are fp32: tap renderer)
        rc_prep = render_prep(P.nbox, P.rec + (L.REC - 1), P.rec + (L.REC - 2), L.ld_rec, c.w.rrec, d->B, L.HW, d->I, d->P, d->align_corners, c.s);
        if (rc_prep != SPAIR_OK && rc_prep != SPAIR_ERR_UNSUPPORTED) return rc_prep;
    }
    // decoder (models.py:474-492)
    const ParamLayout& PL = c.PL;
    const int N = L.N;
    const int per = d->P * d->P * (d->C + 1);
    {
        ProfScope ps(PS_DECODER_FWD, c.s);
        const int b16 = d->dtype == SPAIR_BF16;
        const int K0 = round_up(PL.lin[LIN_DEC0].in, 8), K2 = round_up(PL.lin[LIN_DEC2].in, 8);
        if (b16 && !c.use_chain) TRY(spair_to_bf16(c.w.Za, L.ld_rec, c.w.Za16, L.ld_rec, N, L.ld_rec, c.s));     // the fused chain writes bf16 itself
        if (PL.oc_n) {
            TRY(oc_decoder_fwd(c));
        } else if (c.use_dec_fused) {
            // all three layers + the sprite epilogue in one activation-stationary launch (dec_fused.hip)
            ProfScope p2(PS_DEC2_FWD, c.s);
            TRY(dec_fused_fwd(c.w.Za16, L.ld_rec, c.w.dec_stream, params + PL.lin[LIN_DEC0].b, params + PL.lin[LIN_DEC1].b, params + PL.lin[LIN_DEC2].b,
                              c.w.Hd1, c.w.Hd2, c.w.S, c.w.ld_s, N, d->A, per, d->obj_logit_scale, d->alpha_logit_scale, d->alpha_logit_bias, c.s));
        } else {
        if (b16) {   // hidden activations stored as bf16
            TRY(nt16(c, c.w.Za16, L.ld_rec, c.w.lin_wf[LIN_DEC0], K0, c.w.Hd1, SP_DEC_H1, 1, N, SP_DEC_H1, K0, params + PL.lin[LIN_DEC0].b, nullptr, 0, 1));
            TRY(nt16(c, c.w.Hd1, SP_DEC_H1, c.w.lin_wf[LIN_DEC1], SP_DEC_H1, c.w.Hd2, SP_DEC_H2, 1, N, SP_DEC_H2, SP_DEC_H1,
                     params + PL.lin[LIN_DEC1].b, nullptr, 0, 1));
        } else {
            TRY(fwd_lin(c, LIN_DEC0, c.w.Za, L.ld_rec, c.w.Hd1, SP_DEC_H1, 0, N, params + PL.lin[LIN_DEC0].b, SP_DEC_H1, 1));
            TRY(fwd_lin(c, LIN_DEC1, c.w.Hd1, SP_DEC_H1, c.w.Hd2, SP_DEC_H2, 0, N, params + PL.lin[LIN_DEC1].b, SP_DEC_H2, 1));
        }
        {   // decoder.out with the sprite sigmoid epilogue fused (models.py:485-492)
            ProfScope p2(PS_DEC2_FWD, c.s);
            GemmNT g;
            memset(&g, 0, sizeof(g));
            g.A = c.w.Hd2; g.lda = SP_DEC_H2; g.B = c.w.lin_wf[LIN_DEC2]; g.ldb = K2; g.C = c.w.S; g.ldc = c.w.ld_s; g.M = N; g.N = per; g.K = K2;
            g.bias = params + PL.lin[LIN_DEC2].b; g.sprite_ch = d->C + 1;
            g.c_bf16 = b16 && d->C == 1;      // bf16 step: the sprites leave as 16-bit (grey, alpha) pairs -- half the bytes for the renderer, both
                                              // ways (colour images: fp32 sprites for the generic-channel renderer)
            g.obj_scale = d->obj_logit_scale; g.alpha_scale = d->alpha_logit_scale; g.alpha_bias = d->alpha_logit_bias;
            if (b16) TRY(spair_gemm_nt16_impl(g, false, c.s));
            else TRY(spair_gemm_nt_impl(g, false, d->dtype, c.s));
        }
        }
    }
    // KL + render + loss
    {
        ProfScope ps(PS_RENDER_FWD, c.s);
        // bf16 step: the sampling on the matrix cores from per-object records (render3.hip); every other case on the tap kernels
        int rc = SPAIR_ERR_UNSUPPORTED;
        if (rc_prep == SPAIR_OK) {
            rc = render_fwd_mma(c.w.S, c.w.ld_s, c.w.rrec, x, recon, st->train ? c.w.aux : nullptr, c.w.bce_partial, d->B, L.HW, d->I,
                                d->P, d->align_corners, c.s);
        }
        if (d->C != 1)
            rc = render_fwd_c(c.w.S, c.w.ld_s, P.nbox, P.rec + (L.REC - 1), P.rec + (L.REC - 2), L.ld_rec, x, recon, st->train ? c.w.aux : nullptr,
                              c.w.bce_partial, d->B, L.HW, d->C, d->I, d->P, d->align_corners, c.s);
        else if (rc == SPAIR_ERR_UNSUPPORTED)
            rc = render_fwd(c.w.S, c.w.ld_s, P.nbox, P.rec + (L.REC - 1), P.rec + (L.REC - 2), L.ld_rec, x, recon, st->train ? c.w.aux : nullptr,
                            c.w.bce_partial, d->B, L.HW, d->C, d->I, d->P, d->align_corners, d->dtype == SPAIR_BF16 && !d->obj_conv, c.s);
        TRY(rc);
    }
    if (side && hipStreamWaitEvent(c.s, side->ev[1], 0) != hipSuccess) return SPAIR_ERR_LAUNCH;
    ProfScope psl(PS_LOSS, c.s);
    TRY(loss_finalize(c.w.bce_partial, render_num_blocks(d->B, d->I), c.w.kl_partial, loss_gauss_kl_blocks(L), c.w.klp, d->B,
                      st->kl_scale, d->vae_beta, loss_out,
                      c.use_chain && c.w.chain_sync ? c.w.chain_sync + CHAIN_SYNC_STICKY(d->B, chain_bands(*d)) : nullptr, st->status,
                      st->status_host, c.s));
    return SPAIR_OK;
}

// ---- backward ---------------------------------------------------------------------------------------------
// dX[R, in] = dOut[R, out] . W   (+ relu mask of the layer below)
static int bwd_lin(Ctx& c, int id, int out_total, const float* dOut, int ldo, float* dX, int ldx, int r0, int R, const float* mask, int ldmask) {
    const int K = round_up(out_total, 8);
    return nt(c, dOut + (size_t)r0 * ldo, ldo, c.w.lin_wt[id], K, dX + (size_t)r0 * ldx, ldx, R, c.PL.lin[id].in, K, nullptr,
              mask ? mask + (size_t)r0 * ldmask : nullptr, ldmask, 0);
}
static int wgrad_lin(Ctx& c, int id, const float* dOut, int ldo, const float* In, int ldi, float* grads, int R) {
    const LinSpec& l = c.PL.lin[id];
    return tn(c, dOut, ldo, l.out, In, ldi, l.in, grads + l.w, l.in, R, grads + l.b);
}

// Weight / bias gradients of the 14 per-cell layers in ONE grouped split-K launch (fused-chain path: layer-output gradients and layer
// inputs are both bf16 rows).  38 tiles of <= 128 x 128 at the reference sizes.
static int cells_wgrad_grouped(Ctx& c, float* grads) {
    const CellLayout& L = c.L;
    const CellBufs& P = c.w.cb;
    GemmTN g;
    memset(&g, 0, sizeof(g));
    int nt = 0;
    bool fits = true;
    // dY: bf16 [N][ldo] (columns a0 .. a0+M-1 of it), X: fp32 [N][ldi]
    // input columns [n_lo, n_hi) of layer `id` come from X (bf16 rows, row stride ldi elements), whose column 0 is the layer's input
    // column xcol0.  A tile's operand pointer is rounded down to 8 elements (16 bytes); the columns in front belong to another tile.
    auto add_cols = [&](int id, const float* dY, int ldo, int a0, const float* X_, int ldi, int xcol0, int n_lo, int n_hi, bool with_bias) {
        const LinSpec& l = c.PL.lin[id];
        const __bf16* A = reinterpret_cast<const __bf16*>(dY);
        const __bf16* X = reinterpret_cast<const __bf16*>(X_);
        for (int m0 = 0; m0 < l.out; m0 += 128) {
            const int col = a0 + m0, col_al = col & ~7, skip = col - col_al;          // A tile starts on a 16-byte boundary
            const int ms = std::min(128, l.out - m0);
            const int Ml = std::min(round_up(skip + ms, 8), ldo - col_al);
            int n0 = n_lo;
            while (n0 < n_hi) {
                const int xc = n0 - xcol0, xc_al = xc & ~7, nskip = xc - xc_al;
                const int ns = std::min(128 - nskip, n_hi - n0);
                if (nt >= SPAIR_TN_MAX_TILES || skip + ms > 128) { fits = false; return; }
                GemmTN::Tile& t = g.tile[nt++];
                t.A = A + col_al; t.lda = ldo; t.B = X + xc_al; t.ldb = ldi;
                t.C = grads + l.w + (size_t)m0 * l.in + n0; t.ldc = l.in; t.colsum = (with_bias && n0 == n_lo) ? grads + l.b + m0 : nullptr;
                t.M = Ml; t.N = std::min(round_up(nskip + ns, 8), ldi - xc_al); t.Mstore = ms; t.Nstore = ns; t.m_skip = skip; t.n_skip = nskip;
                if (t.N < nskip + ns) { fits = false; return; }
                n0 += ns;
            }
        }
    };
    auto add = [&](int id, const float* dY, int ldo, int a0, const float* X, int ldi) {
        add_cols(id, dY, ldo, a0, X, ldi, 0, 0, c.PL.lin[id].in, true);
    };
    const int nfc = L.F + L.CTX;      // the [features | context] columns every first layer shares: the fused chain stores them once (Xb)
    add(LIN_BOX0, P.dHb1, SP_LDH, 0, P.Xb, L.ld_xb);
    add(LIN_BOX1, P.dHb2, SP_LDH, 0, P.Hb1, SP_LDH);
    add(LIN_BOXH1, P.dOb, L.ld_ob, 0, P.Hb2, SP_LDH);
    add(LIN_BOXH0, P.dOb, L.ld_ob, L.ob_lat, P.Hb2, SP_LDH);
    // (LIN_ENC0 is not in the group: 14 of its 128 x 128 tiles re-read the 107-MB glimpse rows twice and d He1 seven times; on its own it runs
    //  as 2 x 4 tiles of 128 x 256 with 32 row splits, 4 per XCD -- see below)
    add(LIN_ENC1, P.dHe2, SP_ENC_H2, 0, P.He1, SP_ENC_H1);
    add(LIN_ENC2, P.dOe, L.ld_oe, 0, P.He2, SP_ENC_H2);
    add_cols(LIN_Z0, P.dHz1, SP_LDH, 0, P.Xb, L.ld_xb, 0, 0, nfc, true);
    add_cols(LIN_Z0, P.dHz1, SP_LDH, 0, P.Xz, L.ld_x, 0, nfc, c.PL.lin[LIN_Z0].in, false);
    add(LIN_Z1, P.dHz2, SP_LDH, 0, P.Hz1, SP_LDH);
    add(LIN_ZH1, P.dOz, L.ld_oz, 0, P.Hz2, SP_LDH);
    add(LIN_ZH0, P.dOz, L.ld_oz, L.oz_lat, P.Hz2, SP_LDH);
    add_cols(LIN_OBJ0, P.dHo1, SP_LDH, 0, P.Xb, L.ld_xb, 0, 0, nfc, true);
    add_cols(LIN_OBJ0, P.dHo1, SP_LDH, 0, P.Xo, L.ld_x, 0, nfc, c.PL.lin[LIN_OBJ0].in, false);
    add(LIN_OBJ1, P.dHo2, SP_LDH, 0, P.Ho1, SP_LDH);
    add(LIN_OBJ2, P.dOo, L.ld_oo, 0, P.Ho2, SP_LDH);
    if (!fits) return SPAIR_ERR_UNSUPPORTED;
    g.ngroup = nt; g.R = L.N;
    g.part = c.tn_scratch ? c.tn_scratch : c.w.tn_part; g.part_cap = SPAIR_TN_PART_FLOATS;
    TRY(spair_gemm_tn16_impl(g, false, true, c.s));      // A and B both bf16 rows
    const LinSpec& e0 = c.PL.lin[LIN_ENC0];
    return tn16(c, P.dHe1, SP_ENC_H1, e0.out, P.glimpse, L.ld_gl, e0.in, true, grads + e0.w, e0.in, (int)L.N, grads + e0.b);
}

// The decoder's two small weight gradients (dense1: 256 x 128, dense0: 128 x A) as ONE grouped split-K launch + one reduce pass: as launches of
// their own each took ~28 us + a ~20-35 us reduce for 4.3 / 1.7 GFLOP (prologue, partial tiles and launch latency, not work).
static int decoder_small_wgrad_grouped(Ctx& c, float* grads, long long N) {
    const LinSpec &l1 = c.PL.lin[LIN_DEC1], &l0 = c.PL.lin[LIN_DEC0];
    GemmTN g;
    memset(&g, 0, sizeof(g));
    int nt = 0;
    auto add = [&](const LinSpec& l, const void* dY, int ldo, const void* X, int ldi) {      // whole layers, outputs in tiles of 128 rows, in <= 128
        for (int m0 = 0; m0 < l.out; m0 += 128) {
            GemmTN::Tile& t = g.tile[nt++];
            const int ms = std::min(128, l.out - m0);
            t.A = reinterpret_cast<const __bf16*>(dY) + m0; t.lda = ldo; t.B = X; t.ldb = ldi;
            t.C = grads + l.w + (size_t)m0 * l.in; t.ldc = l.in; t.colsum = grads + l.b + m0;
            t.M = round_up(ms, 8); t.N = std::min(round_up(l.in, 8), ldi); t.Mstore = ms; t.Nstore = l.in; t.m_skip = 0; t.n_skip = 0;
        }
    };
    if (l1.in > 128 || l0.in > 128 || (l1.out & 7) || (l0.out & 7) || ceil_div(l1.out, 128) + ceil_div(l0.out, 128) > SPAIR_TN_MAX_TILES)
        return SPAIR_ERR_UNSUPPORTED;
    add(l1, c.w.dHd2, SP_DEC_H2, c.w.Hd1, SP_DEC_H1);
    add(l0, c.w.dHd1, SP_DEC_H1, c.w.Za16, c.L.ld_rec);
    for (int q = 0; q < nt; ++q) if (g.tile[q].N < g.tile[q].Nstore) return SPAIR_ERR_UNSUPPORTED;
    g.ngroup = nt; g.R = (int)N;
    g.part = c.tn_scratch ? c.tn_scratch : c.w.tn_part; g.part_cap = SPAIR_TN_PART_FLOATS;
    return spair_gemm_tn16_impl(g, false, true, c.s);
}

// Gradient readiness: the backward finishes the three parameter groups in this order -- decoder, per-cell nets, then the backbone together
// with the edge element -- and each group is one contiguous range of the flat gradient buffer (spair_grad_buckets).  ev[i] (a caller-created
// hipEvent_t, or null) is recorded on whichever internal stream completes group i, so a data-parallel caller can start that range's
// all-reduce on its own stream while the remaining backward kernels run.
extern "C" int spair_grad_buckets(const SpairDims* d, int64_t* lo3, int64_t* hi3) {
    if (!d || !lo3 || !hi3) return SPAIR_ERR_SHAPE;
    const ParamLayout P = make_param_layout(*d);
    lo3[0] = P.lin[LIN_DEC0].w; hi3[0] = P.attn_gamma;        // decoder (first ready)
    lo3[1] = P.lin[LIN_BOX0].w; hi3[1] = P.lin[LIN_DEC0].w;   // box / encoder / z / obj nets
    lo3[2] = 0; hi3[2] = P.lin[LIN_BOX0].w;                   // edge element + backbone (last)
    return SPAIR_OK;
}

static int record_ready(void* ev, hipStream_t s) {
    if (ev && hipEventRecord((hipEvent_t)ev, s) != hipSuccess) return SPAIR_ERR_LAUNCH;
    return SPAIR_OK;
}

extern "C" int spair_backward_ev(const SpairDims* d, const SpairStep* st, const float* params, const float* x, const float* eps_box,
                                 const float* eps_attr, const float* eps_depth, const float* u_pres, void* workspace,
                                 const float* grad_loss, float* grads, void* stream, void* ev_decoder, void* ev_cells, void* ev_backbone);

extern "C" int spair_backward(const SpairDims* d, const SpairStep* st, const float* params, const float* x, const float* eps_box,
                              const float* eps_attr, const float* eps_depth, const float* u_pres, void* workspace,
                              const float* grad_loss, float* grads, void* stream) {
    return spair_backward_ev(d, st, params, x, eps_box, eps_attr, eps_depth, u_pres, workspace, grad_loss, grads, stream, nullptr, nullptr, nullptr);
}

extern "C" int spair_backward_ev(const SpairDims* d, const SpairStep* st, const float* params, const float* x, const float* eps_box,
                                 const float* eps_attr, const float* eps_depth, const float* u_pres, void* workspace,
                                 const float* grad_loss, float* grads, void* stream, void* ev_decoder, void* ev_cells, void* ev_backbone) {
    Ctx c;
    TRY(make_ctx(c, d, st, params, x, eps_box, eps_attr, eps_depth, u_pres, workspace, stream));
    if (!grad_loss || !grads) return SPAIR_ERR_SHAPE;
    const CellLayout& L = c.L;
    CellBufs& P = c.w.cb;
    const ParamLayout& PL = c.PL;
    P.gloss = grad_loss;
    const int N = L.N;
    const int per = d->P * d->P * (d->C + 1);
    // renderer -> d logits, d z_where, d z_pres, d z_depth
    const int b16 = d->dtype == SPAIR_BF16;
    {
        ProfScope ps(PS_RENDER_BWD, c.s);
        if (d->C != 1)
            TRY(render_bwd_c(c.w.S, c.w.ld_s, P.nbox, P.rec + (L.REC - 1), P.rec + (L.REC - 2), L.ld_rec, c.w.aux, grad_loss, c.w.dLog, P.g_nbox_r,
                             P.g_pres_r, P.g_depth_r, c.w.ld_s, d->B, L.HW, d->C, d->I, d->P, d->align_corners, d->obj_logit_scale,
                             d->alpha_logit_scale, c.s));
        else
        TRY(render_bwd(c.w.S, c.w.ld_s, P.nbox, P.rec + (L.REC - 1), P.rec + (L.REC - 2), L.ld_rec, c.w.aux, grad_loss, c.w.dLog, P.g_nbox_r,
                       P.g_pres_r, P.g_depth_r, c.w.ld_s, d->B, L.HW, d->C, d->I, d->P, d->align_corners, d->obj_logit_scale,
                       d->alpha_logit_scale, b16 && !d->obj_conv, b16 && !d->obj_conv,
                       b16 && !d->obj_conv && d->C == 1 && render_prep_supported(L.HW, d->I, d->P, d->align_corners) ? c.w.rrec : nullptr, c.s));
    }
    SideStream* side = nullptr;
    if (!(st->flags & 4)) TRY(side_stream(side));
    std::unique_lock<std::mutex> enq_lock;
    if (side) enq_lock = std::unique_lock<std::mutex>(side->enq_mu);
    hipStream_t const main_s = c.s;
    std::function<int()> dec_wgrads;
    if (PL.oc_n) {
        ProfScope ps(PS_DECODER_BWD, c.s);
        TRY(oc_decoder_bwd(c, grads));
        TRY(record_ready(ev_decoder, c.s));
    } else if (b16) {   // decoder, bf16-stored activations and gradients: the data-gradient chain stays on the caller's stream, the three
                 // weight gradients go to the helper stream and overlap with the (latency-bound) per-cell backward chain
        const LinSpec &l2 = PL.lin[LIN_DEC2], &l1 = PL.lin[LIN_DEC1], &l0 = PL.lin[LIN_DEC0];
        float* const dlog_f32 = c.w.dLog;
        if (d->C != 1) {      // the generic-channel renderer left fp32 sprite gradients
            TRY(spair_to_bf16(c.w.dLog, c.w.ld_s, c.w.dLog16, c.w.ld_s, N, c.w.ld_s, c.s));
            c.w.dLog = reinterpret_cast<float*>(c.w.dLog16);
        }
        {
            ProfScope ps(PS_DECODER_BWD, c.s);
            // all three data gradients in one launch (dec_fused_bwd.hip); flags bit 6 / an unsupported shape: three implicit-GEMM launches
            int rc_f = SPAIR_ERR_UNSUPPORTED;
            if (!(st->flags & 64)) {
                ProfScope p2(PS_DEC2_DGRAD, c.s);
                rc_f = dec_fused_bwd(c.w.dLog, c.w.ld_s, c.w.lin_wt[LIN_DEC2], round_up(per, 8), c.w.lin_wt[LIN_DEC1], c.w.lin_wt[LIN_DEC0], c.w.Hd2,
                                     c.w.Hd1, c.w.dHd2, c.w.dHd1, P.g_attr_r, L.ld_rec, N, l0.in, per, c.s);
                if (rc_f != SPAIR_OK && rc_f != SPAIR_ERR_UNSUPPORTED) return rc_f;
            }
            if (rc_f == SPAIR_ERR_UNSUPPORTED) {
            { ProfScope p2(PS_DEC2_DGRAD, c.s);
              TRY(nt16(c, c.w.dLog, c.w.ld_s, c.w.lin_wt[LIN_DEC2], round_up(per, 8), c.w.dHd2, SP_DEC_H2, 1, N, SP_DEC_H2, round_up(per, 8), nullptr,
                       c.w.Hd2, SP_DEC_H2, 0)); }
            TRY(nt16(c, c.w.dHd2, SP_DEC_H2, c.w.lin_wt[LIN_DEC1], SP_DEC_H2, c.w.dHd1, SP_DEC_H1, 1, N, SP_DEC_H1, SP_DEC_H2, nullptr, c.w.Hd1, SP_DEC_H1, 0));
            TRY(nt16(c, c.w.dHd1, SP_DEC_H1, c.w.lin_wt[LIN_DEC0], SP_DEC_H1, P.g_attr_r, L.ld_rec, 0, N, l0.in, SP_DEC_H1, nullptr, nullptr, 0, 0));
            }
        }
        // The decoder's three weight gradients (helper stream).  With the fused chain they are issued BEHIND the chain backward's launch (round 6):
        // its 256 LDS-exclusive workgroups leave them no CU before they retire anyway, but issued in front of it they were eligible the moment
        // the first chain workgroup left -- and took the machine from the 1x1 stack's data gradient, the head of the backbone backward's
        // critical path, which only becomes eligible when the LAST chain workgroup has left.
        float* const dlog_w = c.w.dLog;
        dec_wgrads = [&c, &l2, &l1, &l0, &L, grads, N, dlog_w, ev_decoder]() -> int {
            { ProfScope p2(PS_DEC2_WGRAD, c.s); TRY(tn16(c, dlog_w, c.w.ld_s, l2.out, c.w.Hd2, SP_DEC_H2, l2.in, true, grads + l2.w, l2.in, N, grads + l2.b)); }
            const int rc_g = decoder_small_wgrad_grouped(c, grads, N);
            if (rc_g == SPAIR_ERR_UNSUPPORTED) {
                TRY(tn16(c, c.w.dHd2, SP_DEC_H2, l1.out, c.w.Hd1, SP_DEC_H1, l1.in, true, grads + l1.w, l1.in, N, grads + l1.b));
                TRY(tn16(c, c.w.dHd1, SP_DEC_H1, l0.out, c.w.Za16, L.ld_rec, l0.in, true, grads + l0.w, l0.in, N, grads + l0.b));
            } else if (rc_g != SPAIR_OK) return rc_g;
            return record_ready(ev_decoder, c.s);
        };
        if (!(side && c.use_chain)) {          // no helper stream / per-wavefront launches: where they always were
            if (side) { TRY(stream_link(main_s, side->s, side->ev[0])); c.s = side->s; c.tn_scratch = c.w.tn_part2; }
            TRY(dec_wgrads());
            dec_wgrads = nullptr;
            if (side) { c.s = main_s; c.tn_scratch = nullptr; }
        }
        c.w.dLog = dlog_f32;
    } else {   // decoder
        ProfScope ps(PS_DECODER_BWD, c.s);
        { ProfScope p2(PS_DEC2_WGRAD, c.s); TRY(wgrad_lin(c, LIN_DEC2, c.w.dLog, c.w.ld_s, c.w.Hd2, SP_DEC_H2, grads, N)); }
        { ProfScope p2(PS_DEC2_DGRAD, c.s); TRY(bwd_lin(c, LIN_DEC2, per, c.w.dLog, c.w.ld_s, c.w.dHd2, SP_DEC_H2, 0, N, c.w.Hd2, SP_DEC_H2)); }
        TRY(wgrad_lin(c, LIN_DEC1, c.w.dHd2, SP_DEC_H2, c.w.Hd1, SP_DEC_H1, grads, N));
        TRY(bwd_lin(c, LIN_DEC1, SP_DEC_H2, c.w.dHd2, SP_DEC_H2, c.w.dHd1, SP_DEC_H1, 0, N, c.w.Hd1, SP_DEC_H1));
        TRY(wgrad_lin(c, LIN_DEC0, c.w.dHd1, SP_DEC_H1, c.w.Za, L.ld_rec, grads, N));
        TRY(bwd_lin(c, LIN_DEC0, SP_DEC_H1, c.w.dHd1, SP_DEC_H1, P.g_attr_r, L.ld_rec, 0, N, nullptr, 0));
        TRY(record_ready(ev_decoder, c.s));
    }
    // per-cell chain, reverse wavefront order
    const int ps_cells = prof_begin(PS_CELLS_BWD, c.s);
    ChainArgs chain_args;
    memset(&chain_args, 0, sizeof(chain_args));
    if (c.use_chain) {
        ChainArgs a;
        memset(&a, 0, sizeof(a));
        a.L = L; a.P = P; a.H = c.H;
        for (int i = 0; i < CW_COUNT; ++i) a.wt[i] = reinterpret_cast<const uint4*>(c.w.chain_wt[i]);
        a.w_obj2 = params + PL.lin[LIN_OBJ2].w;
        a.gedge = grads + PL.edge; a.gedge_part = c.w.gedge_part;
        a.x = x; a.I = d->I; a.Pp = d->P; a.ac = d->align_corners;
        a.stamps = (st->flags & 2) ? c.w.stamps : nullptr;
        a.nbands = chain_bands(*d); a.sync = c.w.chain_sync; a.bnd_rec = c.w.bnd_rec; a.bnd_grad = c.w.bnd_grad;
        TRY(chain_bwd(a, c.s));
        chain_args = a;
    } else {
    for (int t = c.T - 1; t >= 0; --t) {
        const int r0 = c.dstart[t] * L.B, R = (c.dstart[t + 1] - c.dstart[t]) * L.B;
        TRY(cells_bwd_pres(L, P, c.H, r0, R, c.s));
        TRY(bwd_lin(c, LIN_OBJ2, 1, P.dOo, L.ld_oo, P.dHo2, SP_LDH, r0, R, P.Ho2, SP_LDH));
        TRY(bwd_lin(c, LIN_OBJ1, SP_H, P.dHo2, SP_LDH, P.dHo1, SP_LDH, r0, R, P.Ho1, SP_LDH));
        TRY(bwd_lin(c, LIN_OBJ0, SP_H, P.dHo1, SP_LDH, P.dXo, L.ld_x, r0, R, nullptr, 0));
        TRY(cells_bwd_depth(L, P, c.H, r0, R, c.s));
        TRY(bwd_lin(c, LIN_ZH1, L.NP + 2, P.dOz, L.ld_oz, P.dHz2, SP_LDH, r0, R, P.Hz2, SP_LDH));
        TRY(bwd_lin(c, LIN_Z1, SP_H, P.dHz2, SP_LDH, P.dHz1, SP_LDH, r0, R, P.Hz1, SP_LDH));
        TRY(bwd_lin(c, LIN_Z0, SP_H, P.dHz1, SP_LDH, P.dXz, L.ld_x, r0, R, nullptr, 0));
        TRY(cells_bwd_attr(L, P, c.H, r0, R, c.s));
        if (PL.oc_n) TRY(oc_encoder_bwd(c, r0, R));
        else {
        TRY(bwd_lin(c, LIN_ENC2, 2 * L.A, P.dOe, L.ld_oe, P.dHe2, SP_ENC_H2, r0, R, P.He2, SP_ENC_H2));
        TRY(bwd_lin(c, LIN_ENC1, SP_ENC_H2, P.dHe2, SP_ENC_H2, P.dHe1, SP_ENC_H1, r0, R, P.He1, SP_ENC_H1));
        TRY(bwd_lin(c, LIN_ENC0, SP_ENC_H1, P.dHe1, SP_ENC_H1, P.dGl, L.ld_gl, r0, R, nullptr, 0));
        }
        TRY(stn_glimpse_bwd(x, P.nbox, L.B, P.dGl, L.ld_gl, P.g_nbox_stn, r0, R, d->C, d->I, d->P, d->align_corners, chain_image_fp16(*d), c.s));
        TRY(cells_bwd_box(L, P, c.H, r0, R, c.s));
        TRY(bwd_lin(c, LIN_BOXH1, L.NP + 8, P.dOb, L.ld_ob, P.dHb2, SP_LDH, r0, R, P.Hb2, SP_LDH));
        TRY(bwd_lin(c, LIN_BOX1, SP_H, P.dHb2, SP_LDH, P.dHb1, SP_LDH, r0, R, P.Hb1, SP_LDH));
        TRY(bwd_lin(c, LIN_BOX0, SP_H, P.dHb1, SP_LDH, P.dXb, L.ld_xb, r0, R, nullptr, 0));
    }
    TRY(cells_dfeat_edge(L, P, grads + PL.edge, c.s));
    }
    prof_end(ps_cells, c.s);
    // the per-cell weight gradients (helper stream) and the backbone backward (caller's stream) both hang off the chain only
    if (side) { TRY(stream_link(main_s, side->s, side->ev[2])); c.s = side->s; }
    if (dec_wgrads) {                          // (deferred: see the decoder block)
        c.tn_scratch = c.w.tn_part2;
        TRY(dec_wgrads());
        c.tn_scratch = nullptr;
    }
    const int ps_wg = prof_begin(PS_CELLS_WGRAD, c.s);
    // weight gradients of the per-cell nets: long-K GEMMs over all N rows
    if (c.use_chain) {
        if (side) c.tn_scratch = c.w.tn_part2;
        TRY(cells_wgrad_grouped(c, grads));
        c.tn_scratch = nullptr;
        TRY(chain_edge_reduce(chain_args, c.s));       // the edge element's gradient: per-sample partials summed in sample order (off the critical path)
    } else {
    TRY(wgrad_lin(c, LIN_BOX0, P.dHb1, SP_LDH, P.Xb, L.ld_xb, grads, N));
    TRY(wgrad_lin(c, LIN_BOX1, P.dHb2, SP_LDH, P.Hb1, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_BOXH1, P.dOb, L.ld_ob, P.Hb2, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_BOXH0, P.dOb + L.ob_lat, L.ld_ob, P.Hb2, SP_LDH, grads, N));
    if (PL.oc_n) TRY(oc_encoder_wgrad(c, grads));
    else {
    TRY(wgrad_lin(c, LIN_ENC0, P.dHe1, SP_ENC_H1, P.glimpse, L.ld_gl, grads, N));
    TRY(wgrad_lin(c, LIN_ENC1, P.dHe2, SP_ENC_H2, P.He1, SP_ENC_H1, grads, N));
    TRY(wgrad_lin(c, LIN_ENC2, P.dOe, L.ld_oe, P.He2, SP_ENC_H2, grads, N));
    }
    TRY(wgrad_lin(c, LIN_Z0, P.dHz1, SP_LDH, P.Xz, L.ld_x, grads, N));
    TRY(wgrad_lin(c, LIN_Z1, P.dHz2, SP_LDH, P.Hz1, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_ZH1, P.dOz, L.ld_oz, P.Hz2, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_ZH0, P.dOz + L.oz_lat, L.ld_oz, P.Hz2, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_OBJ0, P.dHo1, SP_LDH, P.Xo, L.ld_x, grads, N));
    TRY(wgrad_lin(c, LIN_OBJ1, P.dHo2, SP_LDH, P.Ho1, SP_LDH, grads, N));
    TRY(wgrad_lin(c, LIN_OBJ2, P.dOo, L.ld_oo, P.Ho2, SP_LDH, grads, N));
    }
    prof_end(ps_wg, c.s);
    TRY(record_ready(ev_cells, c.s));
    if (side) {
        if (hipEventRecord(side->ev[3], side->s) != hipSuccess) return SPAIR_ERR_LAUNCH;
        c.s = main_s;
    }
    { ProfScope ps(PS_BACKBONE_BWD, c.s); TRY(backbone_bwd(c, grads)); }
    if (side && hipStreamWaitEvent(main_s, side->ev[3], 0) != hipSuccess) return SPAIR_ERR_LAUNCH;      // join
    TRY(record_ready(ev_backbone, main_s));
    return SPAIR_OK;
}

// diagnostic: copy the forward chain kernel's stage stamps (SpairStep.flags bit 1) into a caller buffer of n uint64
extern "C" int spair_chain_stamps(const SpairDims* d, const void* workspace, unsigned long long* out, int n, void* stream) {
    if (!d || !workspace || !out || n > 4096) return SPAIR_ERR_SHAPE;
    TRY(validate(*d));
    const Ws w = carve(*d, const_cast<void*>(workspace));
    if (hipMemcpyAsync(out, w.stamps, sizeof(unsigned long long) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return SPAIR_ERR_LAUNCH;
    return SPAIR_OK;
}

// band split of the fused chain kernels: 1 if a wait for the neighbouring band ever timed out in a launch on this workspace (STICKY: no
// launch clears it; the step's loss and edge-element gradient are NaN from then on -- never seen, the test suite asserts 0), else 0; -1 where
// the chain runs unsplit.  Copies one int to `out` (device).
extern "C" int spair_chain_sync_status(const SpairDims* d, const void* workspace, int* out, void* stream) {
    if (!d || !workspace || !out) return SPAIR_ERR_SHAPE;
    TRY(validate(*d));
    const Ws w = carve(*d, const_cast<void*>(workspace));
    if (!w.chain_sync) return hipMemsetAsync(out, 0xff, sizeof(int), (hipStream_t)stream) == hipSuccess ? SPAIR_OK : SPAIR_ERR_LAUNCH;
    return hipMemcpyAsync(out, w.chain_sync + CHAIN_SYNC_STICKY(d->B, chain_bands(*d)), sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? SPAIR_OK : SPAIR_ERR_LAUNCH;
}
// number of wavefronts the stamping workgroup (sample 0, top band) walks: 3G-2 unsplit, its band's share otherwise
extern "C" int spair_chain_stamp_wavefronts(const SpairDims* d) {
    if (!d) return SPAIR_ERR_SHAPE;
    const int nb = chain_fwd_supported(*d) ? chain_bands(*d) : 1;
    const int hb = (d->G + nb - 1) / nb;
    return 2 * (hb - 1) + d->G;
}

// diagnostic: how the stamp buffer is laid out -- stamps per wavefront of the forward kernel (at offset 0), index of the glimpse-sampling
// interval among its stage intervals (K4), stamps per wavefront of the backward kernel (at offset 2048)
extern "C" int spair_chain_stamp_layout(int* fwd_per_wavefront, int* fwd_glimpse_interval, int* bwd_per_wavefront) {
    if (!fwd_per_wavefront || !fwd_glimpse_interval || !bwd_per_wavefront) return SPAIR_ERR_SHAPE;
    *fwd_per_wavefront = CHAIN_FWD_STAMPS; *fwd_glimpse_interval = CHAIN_FWD_GLIMPSE; *bwd_per_wavefront = CHAIN_BWD_STAMPS;
    return SPAIR_OK;
}

// which: 0 z_attr, 1 z_depth, 2..7 mean of cy,cx,height,width,attr,depth, 8..13 sigma, 14 count-prior p_z
extern "C" int spair_export_map(const SpairDims* d, const void* workspace, int which, float* out, void* stream) {
    if (!d || !workspace || !out) return SPAIR_ERR_SHAPE;
    TRY(validate(*d));
    const CellLayout L = make_cell_layout(*d);
    const Ws w = carve(*d, const_cast<void*>(workspace));
    const CellBufs& P = w.cb;
    hipStream_t s = (hipStream_t)stream;
    const float* src; int ld, col0, ch;
    if (which == 0) { src = P.rec; ld = L.ld_rec; col0 = 4; ch = L.A; }
    else if (which == 1) { src = P.rec; ld = L.ld_rec; col0 = 4 + L.A; ch = 1; }
    else if (which >= 2 && which <= 5) { src = P.stat; ld = SP_LDSTAT; col0 = ST_MU_BOX + (which - 2); ch = 1; }
    else if (which == 6) { src = P.Oe; ld = L.ld_oe; col0 = 0; ch = L.A; }
    else if (which == 7) { src = P.stat; ld = SP_LDSTAT; col0 = ST_MU_DEPTH; ch = 1; }
    else if (which >= 8 && which <= 11) { src = P.stat; ld = SP_LDSTAT; col0 = ST_SD_BOX + (which - 8); ch = 1; }
    else if (which == 12) { src = P.sd_attr; ld = L.ld_rec; col0 = 0; ch = L.A; }
    else if (which == 13) { src = P.stat; ld = SP_LDSTAT; col0 = ST_SD_DEPTH; ch = 1; }
    else if (which == 14) { src = P.stat; ld = SP_LDSTAT; col0 = ST_PZ; ch = 1; }
    else if (which >= 100 && which < 104 || which >= 200 && which < 204) {
        // the last BACKWARD's per-cell latent gradients (tests: the fused chain against the per-wavefront launches, cell by cell)
        const int k = which % 100;
        if (k == 0) { src = P.dOb; ld = L.ld_ob; col0 = L.ob_lat; ch = 8; }
        else if (k == 1) { src = P.dOe; ld = L.ld_oe; col0 = 0; ch = 2 * L.A; }
        else if (k == 2) { src = P.dOz; ld = L.ld_oz; col0 = L.oz_lat; ch = 2; }
        else { src = P.dOo; ld = L.ld_oo; col0 = 0; ch = 1; }
        if (which >= 200) return misc_export16(src, ld, col0, ch, w.cell_h, w.cell_w, d->B, d->G, out, s);
    }
    else return SPAIR_ERR_SHAPE;
    return misc_export(src, ld, col0, ch, w.cell_h, w.cell_w, d->B, d->G, out, s);
}
