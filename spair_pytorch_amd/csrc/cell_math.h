// The per-cell latent transforms of the reference (models.py:322-411, modules.py:167-189), value and
// gradient, as device functions shared by the per-wavefront kernels (cells.hip) and the fused
// persistent chain kernels (chain.hip).  All fp32.
#pragma once
#include "cells.h"

__device__ __forceinline__ float clamp10(float x) { return fminf(fmaxf(x, -10.f), 10.f); }
__device__ __forceinline__ float in10(float x) { return (x >= -10.f && x <= 10.f) ? 1.f : 0.f; }
// value-preserving freeze (models.py:425): f*x + (1-f)*x
__device__ __forceinline__ float freeze_val(float f, float x) { return f * x + (1.f - f) * x; }

__device__ __forceinline__ float kl_gauss(float mu, float sd, float m, float s) {
    const float vr = (sd / s) * (sd / s);
    const float t1 = ((mu - m) / s) * ((mu - m) / s);
    return 0.5f * (vr + t1 - 1.f - logf(vr));
}

struct BoxFwd {
    float mu[4], sd[4];      // post-freeze mean / std of (cy, cx, height, width)
    float box[4];            // (cell_x, cell_y, width, height)
    float nbox[4];           // (xt, yt, xs, ys)
};

// lat = [mean(cy,cx,h,w) | logstd(cy,cx,h,w)], eps in the same order (models.py:322-381)
__device__ __forceinline__ BoxFwd box_forward(const float* lat, const float* eps, const CellHyper& H, int h, int w) {
    BoxFwd o;
    float z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o.mu[k] = freeze_val(H.wheel, lat[k]);
        o.sd[k] = freeze_val(H.wheel, 2.f * sigmoidf_(clamp10(lat[4 + k])));
        z[k] = o.mu[k] + o.sd[k] * eps[k];
    }
    const float ryx = H.max_yx - H.min_yx, rhw = H.max_hw - H.min_hw;
    const float cell_y = ryx * sigmoidf_(clamp10(z[0])) + H.min_yx;
    const float cell_x = ryx * sigmoidf_(clamp10(z[1])) + H.min_yx;
    const float height = rhw * sigmoidf_(clamp10(z[2])) + H.min_hw;
    const float width = rhw * sigmoidf_(clamp10(z[3])) + H.min_hw;
    o.box[0] = cell_x; o.box[1] = cell_y; o.box[2] = width; o.box[3] = height;
    o.nbox[3] = height * H.anchor / H.img;                    // ys
    o.nbox[2] = width * H.anchor / H.img;                     // xs
    o.nbox[1] = H.cell_over_img * (cell_y + (float)h);        // yt
    o.nbox[0] = H.cell_over_img * (cell_x + (float)w);        // xt
    return o;
}

// gn = d(xt,yt,xs,ys), gb = d(cell_x,cell_y,width,height) -> dlat[8] (mean grads | logstd grads)
__device__ __forceinline__ void box_backward(const float* gn, const float* gb, const float* mu, const float* sd, const float* eps,
                                             const float* lat_ls, float zp, float ks, const CellHyper& H, float* dlat) {
    const float ryx = H.max_yx - H.min_yx, rhw = H.max_hw - H.min_hw;
    const float gq[4] = {
        (gb[1] + gn[1] * H.cell_over_img) * ryx,   // cell_y
        (gb[0] + gn[0] * H.cell_over_img) * ryx,   // cell_x
        (gb[3] + gn[3] * H.anchor / H.img) * rhw,  // height
        (gb[2] + gn[2] * H.anchor / H.img) * rhw,  // width
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float z = mu[k] + sd[k] * eps[k];
        const float s = sigmoidf_(clamp10(z));
        const float g_z = gq[k] * s * (1.f - s) * in10(z);
        const float m = H.prior_mean[k], ps = H.prior_std[k];
        const float g_mu = (g_z + ks * zp * (mu[k] - m) / (ps * ps)) * (1.f - H.wheel);
        const float g_sd = (g_z * eps[k] + ks * zp * (sd[k] / (ps * ps) - 1.f / sd[k])) * (1.f - H.wheel);
        const float sl = sigmoidf_(clamp10(lat_ls[k]));
        dlat[k] = g_mu;
        dlat[4 + k] = g_sd * 2.f * sl * (1.f - sl) * in10(lat_ls[k]);
    }
}

__device__ __forceinline__ void attr_forward(float mean, float ls, float eps, float& sd, float& attr) {
    sd = 2.f * sigmoidf_(clamp10(ls));
    attr = mean + sd * eps;
}
__device__ __forceinline__ void attr_backward(float g, float mu, float sd, float ls, float eps, float zp, float ks, const CellHyper& H,
                                              float& d_mean, float& d_ls) {
    const float m = H.prior_mean[4], ps = H.prior_std[4];
    const float g_sd = g * eps + ks * zp * (sd / (ps * ps) - 1.f / sd);
    const float sl = sigmoidf_(clamp10(ls));
    d_mean = g + ks * zp * (mu - m) / (ps * ps);
    d_ls = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
}

__device__ __forceinline__ void depth_forward(float lat_mu, float lat_ls, float eps, const CellHyper& H, float& mu, float& sd, float& depth) {
    mu = freeze_val(H.wheel, lat_mu);
    sd = freeze_val(H.wheel, 2.f * sigmoidf_(clamp10(lat_ls)));
    depth = 4.f * sigmoidf_(clamp10(mu + sd * eps));
}
__device__ __forceinline__ void depth_backward(float g_depth, float mu, float sd, float ls, float eps, float zp, float ks, const CellHyper& H,
                                               float& d_mu, float& d_ls) {
    const float dl = mu + sd * eps;
    const float s = sigmoidf_(clamp10(dl));
    const float g_dl = g_depth * 4.f * s * (1.f - s) * in10(dl);
    const float m = H.prior_mean[5], ps = H.prior_std[5];
    d_mu = (g_dl + ks * zp * (mu - m) / (ps * ps)) * (1.f - H.wheel);
    const float g_sd = (g_dl * eps + ks * zp * (sd / (ps * ps) - 1.f / sd)) * (1.f - H.wheel);
    const float sl = sigmoidf_(clamp10(ls));
    d_ls = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
}

__device__ __forceinline__ float pres_forward(float logit, float u, const CellHyper& H) {
    const float lo = clamp10(freeze_val(H.wheel, logit));
    return sigmoidf_(lo + logf(u + 1e-9f) - logf(1.0f - u + 1e-9f));
}
// g_in = gradient arriving at z_pres from its consumers + renderer; kl = sum of this row's Gaussian KL elements
__device__ __forceinline__ float pres_backward(float g_in, float z, float pz, float kl, float logit, float ks, const CellHyper& H) {
    const float e = 1e-9f;
    // d/dz of z*(log(z+e)-log(pz+e)) + (1-z)*(log(1-z+e)-log(1-pz+e))   (models.py:223-226)
    const float dkl = logf(z + e) - logf(pz + e) + z / (z + e) - logf(1.f - z + e) + logf(1.f - pz + e) - (1.f - z) / (1.f - z + e);
    const float g = g_in + ks * (kl + dkl);
    return g * z * (1.f - z) * in10(freeze_val(H.wheel, logit)) * (1.f - H.wheel);
}
