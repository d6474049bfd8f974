// Spatial-transformer glimpse extraction (K4): stn(x, z_where, [P,P]) of the reference
// (modules.py:216-273 via models.py:383-391) = affine_grid + bilinear grid_sample with BORDER
// padding, and its gradient wrt z_where = (xt, yt, xs, ys)  (x needs no gradient).
// theta = [[xs,0,2xt-1],[0,ys,2yt-1]]; SURVEY.md Appendix A.3 for the coordinate conventions.
#include "cells.h"
#include "stn_math.h"

// one thread per glimpse element; row r reads image b = r % B
// px16: round every pixel through fp16 first -- what the fused per-cell kernel (bf16 mode) sees from its LDS copy of the image
__device__ __forceinline__ float stn_px(const float* img, int o, int px16) { const float v = img[o]; return px16 ? (float)(_Float16)v : v; }

__global__ __launch_bounds__(256) void k_stn_glimpse_fwd(const float* __restrict__ x, const float* __restrict__ nbox, int B,
                                                         float* __restrict__ out, int ld, int r0, int R, int C, int I, int P,
                                                         int ac, int px16) {
    const int per = C * P * P;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * per) return;
    const int i_row = (int)(idx / per);
    const int e = (int)(idx - (long long)i_row * per);
    const int c = e / (P * P), ij = e - c * P * P, i = ij / P, j = ij - i * P;
    const int r = r0 + i_row, b = r % B;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    float ix, iy, mx, my;
    stn_src_coord(nb.z, 2.f * nb.x - 1.f, j, P, I, ac, true, ix, mx);
    stn_src_coord(nb.w, 2.f * nb.y - 1.f, i, P, I, ac, true, iy, my);
    const float* img = x + ((size_t)b * C + c) * I * I;
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    const float wx1 = ix - (float)x0, wy1 = iy - (float)y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool xin = (x0 + 1) < I, yin = (y0 + 1) < I;   // x0,y0 are in [0, I-1] after the border clip
    const float v00 = stn_px(img, y0 * I + x0, px16);
    const float v01 = xin ? stn_px(img, y0 * I + x0 + 1, px16) : 0.f;
    const float v10 = yin ? stn_px(img, (y0 + 1) * I + x0, px16) : 0.f;
    const float v11 = (xin && yin) ? stn_px(img, (y0 + 1) * I + x0 + 1, px16) : 0.f;
    out[(size_t)r * ld + e] = v00 * (wy0 * wx0) + v01 * (wy0 * wx1) + v10 * (wy1 * wx0) + v11 * (wy1 * wx1);
}

// one 256-thread block per row: reduce d(xt,yt,xs,ys) over the C*P*P glimpse elements
__global__ __launch_bounds__(256) void k_stn_glimpse_bwd(const float* __restrict__ x, const float* __restrict__ nbox, int B,
                                                         const float* __restrict__ dgl, int ld, float* __restrict__ dnbox,
                                                         int r0, int C, int I, int P, int ac, int px16) {
    __shared__ float red[4];
    const int r = r0 + blockIdx.x, b = r % B;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    const int per = C * P * P;
    float g_xs = 0.f, g_ys = 0.f, g_tx = 0.f, g_ty = 0.f;
    for (int e = threadIdx.x; e < per; e += blockDim.x) {
        const int c = e / (P * P), ij = e - c * P * P, i = ij / P, j = ij - i * P;
        float ix, iy, mx, my;
        const float X = stn_src_coord(nb.z, 2.f * nb.x - 1.f, j, P, I, ac, true, ix, mx);
        const float Y = stn_src_coord(nb.w, 2.f * nb.y - 1.f, i, P, I, ac, true, iy, my);
        const float* img = x + ((size_t)b * C + c) * I * I;
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const float wx1 = ix - (float)x0, wy1 = iy - (float)y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool xin = (x0 + 1) < I, yin = (y0 + 1) < I;
        const float v00 = stn_px(img, y0 * I + x0, px16);
        const float v01 = xin ? stn_px(img, y0 * I + x0 + 1, px16) : 0.f;
        const float v10 = yin ? stn_px(img, (y0 + 1) * I + x0, px16) : 0.f;
        const float v11 = (xin && yin) ? stn_px(img, (y0 + 1) * I + x0 + 1, px16) : 0.f;
        const float g = dgl[(size_t)r * ld + e];
        const float gix = g * ((v01 - v00) * wy0 + (v11 - v10) * wy1) * mx;   // d/d ix (pixel units) * d ix/d gx
        const float giy = g * ((v10 - v00) * wx0 + (v11 - v01) * wx1) * my;
        g_tx += gix; g_xs += gix * X;
        g_ty += giy; g_ys += giy * Y;
    }
    g_tx = block_reduce_sum_256(g_tx, red);
    g_ty = block_reduce_sum_256(g_ty, red);
    g_xs = block_reduce_sum_256(g_xs, red);
    g_ys = block_reduce_sum_256(g_ys, red);
    if (threadIdx.x == 0) {
        dnbox[(size_t)r * 4 + 0] = 2.f * g_tx;   // tx = 2*xt - 1
        dnbox[(size_t)r * 4 + 1] = 2.f * g_ty;
        dnbox[(size_t)r * 4 + 2] = g_xs;
        dnbox[(size_t)r * 4 + 3] = g_ys;
    }
}

int stn_glimpse_fwd(const float* x, const float* nbox, int B, float* out, int ld, int r0, int R, int C, int I, int P, int ac, int px16,
                    hipStream_t s) {
    if (R <= 0) return SPAIR_ERR_SHAPE;
    const long long total = (long long)R * C * P * P;
    hipLaunchKernelGGL(k_stn_glimpse_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, nbox, B, out, ld, r0, R, C, I, P, ac, px16);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
int stn_glimpse_bwd(const float* x, const float* nbox, int B, const float* dgl, int ld, float* dnbox, int r0, int R, int C, int I,
                    int P, int ac, int px16, hipStream_t s) {
    if (R <= 0) return SPAIR_ERR_SHAPE;
    hipLaunchKernelGGL(k_stn_glimpse_bwd, dim3(R), dim3(256), 0, s, x, nbox, B, dgl, ld, dnbox, r0, C, I, P, ac, px16);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

extern "C" int spair_stn_glimpse_fwd(const float* x, const float* nbox, int B, float* glimpse, int ld_gl, int R, int C, int I,
                                     int P, int align_corners, void* stream) {
    return stn_glimpse_fwd(x, nbox, B, glimpse, ld_gl, 0, R, C, I, P, align_corners, 0, (hipStream_t)stream);
}
extern "C" int spair_stn_glimpse_bwd(const float* x, const float* nbox, int B, const float* dglimpse, int ld_gl, float* dnbox,
                                     int R, int C, int I, int P, int align_corners, void* stream) {
    return stn_glimpse_bwd(x, nbox, B, dglimpse, ld_gl, dnbox, 0, R, C, I, P, align_corners, 0, (hipStream_t)stream);
}
