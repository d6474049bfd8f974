// Dimensions, parameter layout and workspace layout of the SPAIR step (host + device view).
// Single source of truth for both sides of the C ABI: Python asks spair_param_layout() /
// spair_workspace_bytes() instead of re-deriving anything.
#pragma once
#include "common.h"
#include "spair_hip.h"

// ---- fixed network widths (reference: models.py:145-165, config.py:6) ----------------------
#define SP_H 100        // hidden width of box / z / obj MLPs
#define SP_LDH 104      // its padded leading dimension
#define SP_ENC_H1 256
#define SP_ENC_H2 128
#define SP_DEC_H1 128
#define SP_DEC_H2 256
#define SP_MAX_CONV 8

// Row r = cprime * B + b, where cprime enumerates cells in dependency-wavefront order
// (t = 2h + w ascending, then h ascending): every diagonal is a contiguous row range.
struct CellLayout {
    int B, G, HW, N;            // N = B*HW
    int F, A, NP, REC;          // backbone features, attrs, passthrough, record width (4+A+2)
    int LB, NB;                 // N_LOOKBACK and its 2*LB*(LB+1) context neighbours
    int CTX;                    // NB*REC
    // column offsets inside the X buffers (reference concat order, models.py:76,88,100)
    int x_ctx, x_pass, x_box, x_attr, x_depth;
    int ld_xb, ld_x;            // leading dims of Xb (F+CTX padded) and Xz/Xo
    int ld_ob, ob_lat;          // box head output: [pass NP | lat 8]
    int ld_oz, oz_lat;          // z head output:   [pass NP | lat 2]
    int ld_oo;                  // obj output (logit at col 0)
    int ld_oe;                  // encoder output [mean A | logstd A]
    int ld_rec;                 // record [box4 | attr A | depth | pres]
    int glimpse;                // C*P*P
    int ld_gl;
};

static inline CellLayout make_cell_layout(const SpairDims& d) {
    CellLayout L;
    L.B = d.B; L.G = d.G; L.HW = d.G * d.G; L.N = d.B * L.HW;
    L.F = d.F; L.A = d.A; L.NP = d.NP; L.REC = 4 + d.A + 2;
    L.LB = d.lookback > 0 ? d.lookback : 1; L.NB = 2 * L.LB * (L.LB + 1); L.CTX = L.NB * L.REC;
    L.x_ctx = L.F; L.x_pass = L.F + L.CTX; L.x_box = L.x_pass + L.NP; L.x_attr = L.x_box + 4;
    L.x_depth = L.x_attr + L.A;
    L.ld_xb = round_up(L.F + L.CTX, 8);
    L.ld_x = round_up(L.x_depth + 1, 8);
    L.ob_lat = L.NP; L.ld_ob = round_up(L.NP + 8, 8);
    L.oz_lat = L.NP; L.ld_oz = round_up(L.NP + 2, 8);
    L.ld_oo = 8;
    L.ld_oe = round_up(2 * L.A, 8);
    L.ld_rec = round_up(L.REC, 8);
    L.glimpse = d.C * d.P * d.P;
    L.ld_gl = round_up(L.glimpse, 8);
    return L;
}

// ---- parameters (flat fp32 buffer; every tensor starts 16-byte aligned) --------------------
enum LinId {
    LIN_BOX0, LIN_BOX1, LIN_BOXH0, LIN_BOXH1,
    LIN_ENC0, LIN_ENC1, LIN_ENC2,
    LIN_Z0, LIN_Z1, LIN_ZH0, LIN_ZH1,
    LIN_OBJ0, LIN_OBJ1, LIN_OBJ2,
    LIN_DEC0, LIN_DEC1, LIN_DEC2,
    LIN_COUNT
};

struct LinSpec { int in, out; int64_t w, b; };
struct ConvSpec { int cin, cout, k, s; int hin, hout; int64_t w, b; };  // square images

struct ParamLayout {
    int64_t edge;
    int n_conv;                       // including conv_out as the last entry
    ConvSpec conv[SP_MAX_CONV + 1];
    LinSpec lin[LIN_COUNT];
    // convolutional object encoder / decoder variant (SpairDims.obj_conv): oc_n > 0 -- LIN_ENC0/1 and LIN_DEC1/2 are empty, LIN_ENC2 is
    // Linear(oc_flat -> 2A), LIN_DEC0 is Linear(A -> oc_flat); oc_dec[i] mirrors oc_enc[oc_n-1-i] (weights [cin][cout][k][k])
    int oc_n, oc_flat;
    ConvSpec oc_enc[4], oc_dec[4];
    int64_t attn_gamma, attn_q_w, attn_q_b, attn_k_w, attn_k_b, attn_v_w, attn_v_b;
    int64_t total;
};

static inline int64_t pl_take(int64_t& cur, int64_t n) {
    int64_t o = cur;
    cur += (n + 3) / 4 * 4;
    return o;
}

static inline ParamLayout make_param_layout(const SpairDims& d) {
    ParamLayout P;
    int64_t cur = 0;
    const CellLayout L = make_cell_layout(d);
    P.edge = pl_take(cur, L.REC);
    P.n_conv = d.n_conv + 1;
    int cin = d.C, h = d.I + d.pad_pre + d.pad_post;
    for (int i = 0; i < d.n_conv; ++i) {
        ConvSpec& c = P.conv[i];
        c.cin = cin; c.cout = d.conv_c[i]; c.k = d.conv_k[i]; c.s = d.conv_s[i];
        c.hin = h; c.hout = (h - c.k) / c.s + 1;
        c.w = pl_take(cur, (int64_t)c.cout * cin * c.k * c.k);
        c.b = pl_take(cur, c.cout);
        cin = c.cout; h = c.hout;
    }
    {
        ConvSpec& c = P.conv[d.n_conv];
        c.cin = cin; c.cout = d.F; c.k = 1; c.s = 1; c.hin = h; c.hout = h;
        c.w = pl_take(cur, (int64_t)c.cout * cin);
        c.b = pl_take(cur, c.cout);
    }
    auto lin = [&](int id, int in, int out) {
        P.lin[id].in = in; P.lin[id].out = out;
        P.lin[id].w = pl_take(cur, (int64_t)in * out);
        P.lin[id].b = pl_take(cur, out);
    };
    lin(LIN_BOX0, L.F + L.CTX, SP_H); lin(LIN_BOX1, SP_H, SP_H); lin(LIN_BOXH0, SP_H, 8); lin(LIN_BOXH1, SP_H, L.NP);
    P.oc_n = 0; P.oc_flat = 0;
    if (d.obj_conv) {
        P.oc_n = d.oc_n;
        int ci = d.C, hh = d.P;
        for (int i = 0; i < d.oc_n; ++i) {
            ConvSpec& c = P.oc_enc[i];
            c.cin = ci; c.cout = d.oc_c[i]; c.k = d.oc_k[i]; c.s = d.oc_s[i];
            c.hin = hh; c.hout = (hh - c.k) / c.s + 1;
            c.w = pl_take(cur, (int64_t)c.cout * ci * c.k * c.k);
            c.b = pl_take(cur, c.cout);
            ci = c.cout; hh = c.hout;
        }
        P.oc_flat = ci * hh * hh;
        lin(LIN_ENC0, 0, 0); lin(LIN_ENC1, 0, 0); lin(LIN_ENC2, P.oc_flat, 2 * L.A);
    } else {
    lin(LIN_ENC0, L.glimpse, SP_ENC_H1); lin(LIN_ENC1, SP_ENC_H1, SP_ENC_H2); lin(LIN_ENC2, SP_ENC_H2, 2 * L.A);
    }
    const int zin = L.x_depth;  // F + CTX + NP + 4 + A
    lin(LIN_Z0, zin, SP_H); lin(LIN_Z1, SP_H, SP_H); lin(LIN_ZH0, SP_H, 2); lin(LIN_ZH1, SP_H, L.NP);
    lin(LIN_OBJ0, zin + 1, SP_H); lin(LIN_OBJ1, SP_H, SP_H); lin(LIN_OBJ2, SP_H, 1);
    if (d.obj_conv) {
        lin(LIN_DEC0, L.A, P.oc_flat); lin(LIN_DEC1, 0, 0); lin(LIN_DEC2, 0, 0);
        for (int i = 0; i < d.oc_n; ++i) {
            const ConvSpec& e = P.oc_enc[d.oc_n - 1 - i];
            ConvSpec& c = P.oc_dec[i];
            c.cin = e.cout; c.cout = (d.oc_n - 1 - i == 0) ? d.C + 1 : e.cin; c.k = e.k; c.s = e.s;
            c.hin = e.hout; c.hout = e.hin;
            c.w = pl_take(cur, (int64_t)c.cin * c.cout * c.k * c.k);
            c.b = pl_take(cur, c.cout);
        }
    } else {
    lin(LIN_DEC0, L.A, SP_DEC_H1); lin(LIN_DEC1, SP_DEC_H1, SP_DEC_H2);
    lin(LIN_DEC2, SP_DEC_H2, d.P * d.P * (d.C + 1));
    }
    const int ad = 4 + d.A + 1;  // Self_Attn(55) -- dead in the reference, kept for state_dict parity
    P.attn_gamma = pl_take(cur, 1);
    P.attn_q_w = pl_take(cur, (int64_t)(ad / 8) * ad); P.attn_q_b = pl_take(cur, ad / 8);
    P.attn_k_w = pl_take(cur, (int64_t)(ad / 8) * ad); P.attn_k_b = pl_take(cur, ad / 8);
    P.attn_v_w = pl_take(cur, (int64_t)ad * ad); P.attn_v_b = pl_take(cur, ad);
    P.total = cur;
    return P;
}
