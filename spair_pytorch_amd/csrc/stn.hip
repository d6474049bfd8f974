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

// ---------------------------------------------------------------------------------------------
// stn(image, z_where, [I,I], inverse=True) on its own (modules.py:256-269): every sprite [C,P,P] is placed on an [C,I,I]
// canvas through the inverse affine (closed form [1/xs, -t/xs]; the reference inverts a 3x3 by LU, <= 5e-5 away),
// bilinear, ZEROS padding.  This materialises [N,C,I,I] -- the training step never does (the renderer fuses it,
// render2.hip); the entry points exist for callers of the reference's helper (test_renderer.py / notebook style) and
// are not performance-critical: the backward scatters with float atomics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stn_inverse_fwd(const float* __restrict__ sp, const float* __restrict__ nbox, float* __restrict__ out,
                                                         long long total, int C, int P, int I, int ac) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % I), py = (int)((idx / I) % I);
    const long long nc = idx / ((long long)I * I);
    const int c = (int)(nc % C);
    const long long n = nc / C;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + n * 4);
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    float sx, sy, mx, my;
    stn_src_coord(1.f / nb.z, -tx / nb.z, px, I, P, ac, false, sx, mx);
    stn_src_coord(1.f / nb.w, -ty / nb.w, py, I, P, ac, false, sy, my);
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* s = sp + ((size_t)n * C + c) * P * P;
    auto tap = [&](int yy, int xx) -> float { return (yy >= 0 && yy < P && xx >= 0 && xx < P) ? s[yy * P + xx] : 0.f; };
    float v = 0.f;
    if (sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P)
        v = tap(y0, x0) * (wy0 * wx0) + tap(y0, x0 + 1) * (wy0 * wx1) + tap(y0 + 1, x0) * (wy1 * wx0) + tap(y0 + 1, x0 + 1) * (wy1 * wx1);
    out[idx] = v;
}

// dsprite (zero-filled by the caller) and dnbox [N,4] (zero-filled) accumulate by atomics
__global__ __launch_bounds__(256) void k_stn_inverse_bwd(const float* __restrict__ sp, const float* __restrict__ nbox, const float* __restrict__ gout,
                                                         float* __restrict__ dsp, float* __restrict__ dnbox, long long total, int C, int P, int I,
                                                         int ac) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int px = (int)(idx % I), py = (int)((idx / I) % I);
    const long long nc = idx / ((long long)I * I);
    const int c = (int)(nc % C);
    const long long n = nc / C;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + n * 4);
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    const float ax = 1.f / nb.z, ay = 1.f / nb.w;
    float sx, sy, mx, my;
    const float X = stn_src_coord(ax, -tx * ax, px, I, P, ac, false, sx, mx);
    const float Y = stn_src_coord(ay, -ty * ay, py, I, P, ac, false, sy, my);
    if (!(sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P)) return;
    const float g = gout[idx];
    if (g == 0.f) return;
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float* s = sp + ((size_t)n * C + c) * P * P;
    float* d = dsp + ((size_t)n * C + c) * P * P;
    auto ok = [&](int yy, int xx) { return yy >= 0 && yy < P && xx >= 0 && xx < P; };
    const float v00 = ok(y0, x0) ? s[y0 * P + x0] : 0.f, v01 = ok(y0, x0 + 1) ? s[y0 * P + x0 + 1] : 0.f;
    const float v10 = ok(y0 + 1, x0) ? s[(y0 + 1) * P + x0] : 0.f, v11 = ok(y0 + 1, x0 + 1) ? s[(y0 + 1) * P + x0 + 1] : 0.f;
    if (ok(y0, x0)) atomicAdd(&d[y0 * P + x0], g * wy0 * wx0);
    if (ok(y0, x0 + 1)) atomicAdd(&d[y0 * P + x0 + 1], g * wy0 * wx1);
    if (ok(y0 + 1, x0)) atomicAdd(&d[(y0 + 1) * P + x0], g * wy1 * wx0);
    if (ok(y0 + 1, x0 + 1)) atomicAdd(&d[(y0 + 1) * P + x0 + 1], g * wy1 * wx1);
    // d/d(source coords) -> d(scale', shift') of the INVERSE affine -> d(xt, yt, xs, ys):  g_norm = ax*base - tx*ax
    const float g_sx = g * ((v01 - v00) * wy0 + (v11 - v10) * wy1) * mx;      // wrt the normalised source x
    const float g_sy = g * ((v10 - v00) * wx0 + (v11 - v01) * wx1) * my;
    // d g/d tx = -ax, d g/d xs = -(X - tx) * ax^2 = -g_n * ax  (X: base output coordinate)
    atomicAdd(&dnbox[n * 4 + 0], 2.f * (-ax) * g_sx);
    atomicAdd(&dnbox[n * 4 + 1], 2.f * (-ay) * g_sy);
    atomicAdd(&dnbox[n * 4 + 2], -(X - tx) * ax * ax * g_sx);
    atomicAdd(&dnbox[n * 4 + 3], -(Y - ty) * ay * ay * g_sy);
}

extern "C" int spair_stn_inverse_fwd(const float* sprites, const float* nbox, float* out, int N, int C, int P, int I, int align_corners,
                                     void* stream) {
    if (N <= 0 || C <= 0 || P <= 0 || I <= 0) return SPAIR_ERR_SHAPE;
    const long long total = (long long)N * C * I * I;
    hipLaunchKernelGGL(k_stn_inverse_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sprites, nbox, out, total, C,
                       P, I, align_corners);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
extern "C" int spair_stn_inverse_bwd(const float* sprites, const float* nbox, const float* grad_out, float* dsprites, float* dnbox, int N,
                                     int C, int P, int I, int align_corners, void* stream) {
    if (N <= 0 || C <= 0 || P <= 0 || I <= 0) return SPAIR_ERR_SHAPE;
    const long long total = (long long)N * C * I * I;
    hipLaunchKernelGGL(k_stn_inverse_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sprites, nbox, grad_out,
                       dsprites, dnbox, total, C, P, I, align_corners);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
