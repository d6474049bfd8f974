// K6: the SPAIR renderer (reference: models.py:485-547, stn(inverse=True) modules.py:256-269).
//
// The reference materialises [N,3,I,I] (12.9 GB at B=256, G=16).  Here nothing of that size ever
// exists:
//  * forward is PIXEL-centric: one workgroup per (sample, 16x16 output tile) culls the HW objects
//    to the ones whose zero-padded footprint touches the tile (ballot-compacted into LDS), then each
//    thread gathers the 4 bilinear taps of (grey, alpha) per surviving object straight from the
//    [N,P,P,2] sprite array and accumulates  sum_k g a (m+1e-9) / sum_k (m+1e-9)  in registers;
//    importance m is rebuilt per tap from alpha (max(alpha*pres*depth, 0.01)), so only grey+alpha
//    are ever read.  BCE and everything backward needs per pixel are produced in the same pass.
//  * backward is OBJECT-centric: one workgroup per sprite walks that object's pixel footprint,
//    scatters tap gradients into an LDS copy of the sprite with ds_add_f32 (no global atomics:
//    each sprite gradient is written exactly once, coalesced) and reduces d(z_where, pres, depth).
// Workgroup -> (sample, tile) mapping is XCD-aware: all tiles of one sample run on one XCD so the
// sample's sprites (HW * 6.3 KB) are fetched from HBM once and then hit in that XCD's L2.
#include "cells.h"
#include "stn_math.h"

#define RT 16            // output tile side
#define RCH 256          // objects culled per pass (= threads per block)

struct Cand {
    float ax, bx, ay, by, pres, depth;
    int row;
};

// sigmoid epilogue of the decoder (models.py:485-492): in place logits -> (grey, alpha)
__global__ __launch_bounds__(256) void k_sprite_act(float* __restrict__ S, int ld, long long total, int per, int CH, float obj_scale,
                                                    float alpha_scale, float alpha_bias) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long r = idx / per;
    const int e = (int)(idx - r * per);
    float* p = S + r * ld + e;
    const float l = *p;
    const bool is_alpha = (e % CH) == CH - 1;
    const float t = is_alpha ? l * alpha_scale + alpha_bias : l * obj_scale;
    *p = 1.f / (expf(-t) + 1.f);   // analytical sigmoid (modules.py:186-187)
}

__device__ __forceinline__ float src_of(float a, float b, int j, int nout, int nsrc, int ac) {
    float c, m;
    stn_src_coord(a, b, j, nout, nsrc, ac, false, c, m);
    return c;
}

__global__ __launch_bounds__(256) void k_render_fwd(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                    const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                    const float* __restrict__ x, float* __restrict__ recon, float4* __restrict__ aux,
                                                    float* __restrict__ bce_partial, int B, int HW, int I, int P, int ac) {
    __shared__ Cand cand[RCH];
    __shared__ int ncand_sh;
    __shared__ int wave_cnt[4];
    __shared__ float red[4];
    const int tiles_x = (I + RT - 1) / RT, tiles = tiles_x * tiles_x;
    int b, tile;
    if ((B & 7) == 0) {   // XCD-aware: blocks id, id+8, ... share an XCD (round-robin dispatch)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = (j / tiles) * 8 + xcd;
        tile = j % tiles;
    } else {
        b = blockIdx.x / tiles;
        tile = blockIdx.x % tiles;
    }
    const int tx0 = (tile % tiles_x) * RT, ty0 = (tile / tiles_x) * RT;
    const int lx = threadIdx.x & (RT - 1), ly = threadIdx.x >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < I && py < I;
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    float num = 0.f, den = 0.f;
    for (int k0 = 0; k0 < HW; k0 += RCH) {
        // ---- cull RCH objects against this tile
        const int k = k0 + threadIdx.x;
        bool hit = false;
        Cand c;
        if (k < HW) {
            const int r = k * B + b;
            const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
            const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
            c.ax = 1.f / nb.z; c.bx = -tx / nb.z; c.ay = 1.f / nb.w; c.by = -ty / nb.w;
            c.pres = pres[(size_t)r * ld_pd]; c.depth = depth[(size_t)r * ld_pd]; c.row = r;
            // the zero-padded sprite is non-zero for source coords in (-1, P)
            hit = src_of(c.ax, c.bx, tx1, I, P, ac) > -1.f && src_of(c.ax, c.bx, tx0, I, P, ac) < (float)P &&
                  src_of(c.ay, c.by, ty1, I, P, ac) > -1.f && src_of(c.ay, c.by, ty0, I, P, ac) < (float)P;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wave_cnt[w];
        if (hit) cand[base + __popcll(bal & ((1ull << lane) - 1ull))] = c;
        if (threadIdx.x == 0) ncand_sh = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
        const int nc = ncand_sh;
        // ---- accumulate the surviving objects at this thread's pixel
        if (inside) {
            for (int ci = 0; ci < nc; ++ci) {
                const Cand q = cand[ci];
                const float sx = src_of(q.ax, q.bx, px, I, P, ac);
                const float sy = src_of(q.ay, q.by, py, I, P, ac);
                if (!(sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P)) continue;
                const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
                const float wx1 = sx - (float)x0, wy1 = sy - (float)y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const float* sp = S + (size_t)q.row * ld_s;
                const float pd = q.pres * q.depth;
                float g = 0.f, a = 0.f, m = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                    if (yy < 0 || yy >= P || xx < 0 || xx >= P) continue;
                    const float w = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                    const float2 v = *reinterpret_cast<const float2*>(sp + (yy * P + xx) * 2);
                    g += w * v.x;
                    a += w * (v.y * q.pres);
                    m += w * fmaxf(v.y * pd, 0.01f);
                }
                num += g * a * (m + 1e-9f);
                den += m;
            }
        }
        __syncthreads();
    }
    float bce = 0.f;
    if (inside) {
        const float D = den + (float)HW * 1e-9f;   // every object adds 1e-9 (models.py:527)
        const float invD = 1.f / D;
        const float pre = num * invD;
        const float r = fminf(fmaxf(pre, 0.f), 1.f);
        const size_t pi = ((size_t)b * I + py) * I + px;
        const float xv = x[pi];
        recon[pi] = r;
        // torch BCE: log clamped at -100; backward denominator max(r(1-r), 1e-12)
        bce = -(xv * fmaxf(logf(r), -100.f) + (1.f - xv) * fmaxf(logf(1.f - r), -100.f));
        if (aux) {
            const float gr = (pre >= 0.f && pre <= 1.f) ? (r - xv) / fmaxf(r * (1.f - r), 1e-12f) : 0.f;
            aux[pi] = make_float4(gr, invD, pre, 0.f);
        }
    }
    bce = block_reduce_sum_256(bce, red);
    if (threadIdx.x == 0) bce_partial[blockIdx.x] = bce;
}

// ---------------------------------------------------------------------------------------------
// backward: one WAVE per object (4 objects of the same sample per workgroup).  Each wave keeps its sprite and the
// sprite gradient in LDS, walks the object's pixel footprint, scatters tap gradients with ds_add_f32, reduces
// d(z_where, pres, depth) with wave shuffles (no block barriers in the hot path) and writes dlogits once.
// Workgroup -> sample mapping is XCD-aware: every object of sample b runs on XCD b%8, so the per-pixel aux map of a
// sample (I*I*16 B) is only ever cached in one L2.
// ---------------------------------------------------------------------------------------------
#define RB_WAVES 4
__global__ __launch_bounds__(64 * RB_WAVES) void k_render_bwd(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                              const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                              const float4* __restrict__ aux, const float* __restrict__ gloss,
                                                              float* __restrict__ dlogits, float* __restrict__ dnbox, float* __restrict__ dpres,
                                                              float* __restrict__ ddepth, int ld_g, int B, int HW, int I, int P, int ac,
                                                              float obj_scale, float alpha_scale, int g_bf16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int PP2 = P * P * 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Ssh = sm + (size_t)wave * 2 * PP2;     // sprite (grey, alpha)
    float* dSh = Ssh + PP2;                        // its gradient
    // (sample, object) of this wave
    const int kgroups = (HW + RB_WAVES - 1) / RB_WAVES;
    int b, kg;
    if ((B & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = (j % (B >> 3)) * 8 + xcd;
        kg = j / (B >> 3);
    } else {
        b = blockIdx.x % B;
        kg = blockIdx.x / B;
    }
    const int k = kg * RB_WAVES + wave;
    const bool live = k < HW && kg < kgroups;
    const int r = live ? k * B + b : 0;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    const float ax = 1.f / nb.z, bx = -tx / nb.z, ay = 1.f / nb.w, by = -ty / nb.w;
    const float pr = pres[(size_t)r * ld_pd], dp = depth[(size_t)r * ld_pd], pd = pr * dp;
    const float gl = *gloss;
    if (live) {
        for (int e = lane * 4; e < PP2; e += 256) {
            *reinterpret_cast<float4*>(&Ssh[e]) = *reinterpret_cast<const float4*>(S + (size_t)r * ld_s + e);
            *reinterpret_cast<float4*>(&dSh[e]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    float g_tx = 0.f, g_ty = 0.f, g_xs = 0.f, g_ys = 0.f, g_pr = 0.f, g_dp = 0.f;
    if (live) {
        // pixel footprint: source coord is affine in the pixel index; widen by 2 and test exactly below
        const float sx0 = src_of(ax, bx, 0, I, P, ac), sxa = src_of(ax, bx, 1, I, P, ac) - sx0;
        const float sy0 = src_of(ay, by, 0, I, P, ac), sya = src_of(ay, by, 1, I, P, ac) - sy0;
        int X0 = (int)floorf((-1.f - sx0) / sxa) - 2, X1 = (int)ceilf(((float)P - sx0) / sxa) + 2;
        int Y0 = (int)floorf((-1.f - sy0) / sya) - 2, Y1 = (int)ceilf(((float)P - sy0) / sya) + 2;
        X0 = max(X0, 0); Y0 = max(Y0, 0); X1 = min(X1, I - 1); Y1 = min(Y1, I - 1);
        const int fw = X1 - X0 + 1, fh = Y1 - Y0 + 1;
        const float mult = ac ? 0.5f * (float)(P - 1) : 0.5f * (float)P;
        const int npx = (fw > 0 && fh > 0) ? fw * fh : 0;
        for (int idx = lane; idx < npx; idx += 64) {
            const int py = Y0 + idx / fw, px = X0 + idx % fw;
            const float sx = src_of(ax, bx, px, I, P, ac), sy = src_of(ay, by, py, I, P, ac);
            if (!(sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P)) continue;
            const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
            const float wx1 = sx - (float)x0, wy1 = sy - (float)y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            float G[4], A0[4], Mt[4], W[4];
            bool ok[4];
            float g = 0.f, a = 0.f, m = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                ok[t] = !(yy < 0 || yy >= P || xx < 0 || xx >= P);
                W[t] = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                const float2 v = ok[t] ? *reinterpret_cast<const float2*>(&Ssh[(yy * P + xx) * 2]) : make_float2(0.f, 0.f);
                G[t] = v.x;
                A0[t] = v.y;
                Mt[t] = ok[t] ? fmaxf(A0[t] * pd, 0.01f) : 0.f;
                g += W[t] * G[t];
                a += W[t] * (A0[t] * pr);
                m += W[t] * Mt[t];
            }
            const float4 av = aux[((size_t)b * I + py) * I + px];   // (dBCE/dpre, 1/D, pre, -)
            const float go = av.x * gl, invD = av.y, pre = av.z;
            const float Dm = m + 1e-9f;
            const float d_g = go * a * Dm * invD;
            const float d_a = go * g * Dm * invD;
            const float d_m = go * (a * g - pre) * invD;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!ok[t]) continue;
                const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                const bool act = (A0[t] * pd) >= 0.01f;
                atomicAdd(&dSh[(yy * P + xx) * 2], W[t] * d_g);
                atomicAdd(&dSh[(yy * P + xx) * 2 + 1], W[t] * (d_a * pr + (act ? d_m * pd : 0.f)));
                g_pr += W[t] * (d_a * A0[t] + (act ? d_m * A0[t] * dp : 0.f));
                g_dp += act ? W[t] * d_m * A0[t] * pr : 0.f;
            }
            // gradient wrt the source coordinate (taps outside the sprite are zeros)
            const float At[4] = {A0[0] * pr, A0[1] * pr, A0[2] * pr, A0[3] * pr};
            const float dgx = (G[1] - G[0]) * wy0 + (G[3] - G[2]) * wy1, dgy = (G[2] - G[0]) * wx0 + (G[3] - G[1]) * wx1;
            const float dax = (At[1] - At[0]) * wy0 + (At[3] - At[2]) * wy1, day = (At[2] - At[0]) * wx0 + (At[3] - At[1]) * wx1;
            const float dmx = (Mt[1] - Mt[0]) * wy0 + (Mt[3] - Mt[2]) * wy1, dmy = (Mt[2] - Mt[0]) * wx0 + (Mt[3] - Mt[1]) * wx1;
            const float g_gx = (d_g * dgx + d_a * dax + d_m * dmx) * mult;   // d/d(normalised source x)
            const float g_gy = (d_g * dgy + d_a * day + d_m * dmy) * mult;
            const float gxn = ax * stn_base(px, I, ac) + bx, gyn = ay * stn_base(py, I, ac) + by;
            g_tx += -g_gx * ax; g_xs += -g_gx * gxn * ax;     // g = (X - t)/s
            g_ty += -g_gy * ay; g_ys += -g_gy * gyn * ay;
        }
    }
    g_tx = wave_reduce_sum(g_tx); g_ty = wave_reduce_sum(g_ty);
    g_xs = wave_reduce_sum(g_xs); g_ys = wave_reduce_sum(g_ys);
    g_pr = wave_reduce_sum(g_pr); g_dp = wave_reduce_sum(g_dp);
    if (live && lane == 0) {
        *reinterpret_cast<float4*>(dnbox + (size_t)r * 4) = make_float4(2.f * g_tx, 2.f * g_ty, g_xs, g_ys);
        dpres[r] = g_pr;
        ddepth[r] = g_dp;
    }
    __syncthreads();
    // through the analytical sigmoid and the logit scales (models.py:485-492)
    if (live) {
        for (int e = lane * 4; e < PP2; e += 256) {
            const float4 sv = *reinterpret_cast<const float4*>(&Ssh[e]);
            const float4 dv = *reinterpret_cast<const float4*>(&dSh[e]);
            float4 o;
            o.x = dv.x * sv.x * (1.f - sv.x) * obj_scale;
            o.y = dv.y * sv.y * (1.f - sv.y) * alpha_scale;
            o.z = dv.z * sv.z * (1.f - sv.z) * obj_scale;
            o.w = dv.w * sv.w * (1.f - sv.w) * alpha_scale;
            if (g_bf16) {
                bf16x4 ob;
                ob[0] = (__bf16)o.x; ob[1] = (__bf16)o.y; ob[2] = (__bf16)o.z; ob[3] = (__bf16)o.w;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dlogits) + (size_t)r * ld_g + e) = ob;
            } else {
                *reinterpret_cast<float4*>(dlogits + (size_t)r * ld_g + e) = o;
            }
        }
    }
}

int render_sprite_act(float* S, int ld, int N, int per, int CH, float obj_scale, float alpha_scale, float alpha_bias, hipStream_t s) {
    const long long total = (long long)N * per;
    hipLaunchKernelGGL(k_sprite_act, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, S, ld, total, per, CH, obj_scale,
                       alpha_scale, alpha_bias);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int render_num_blocks(int B, int I) {
    const int t = (I + RT - 1) / RT;
    return B * t * t;
}

int render_fwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x,
               float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int ac, hipStream_t s) {
    if (C != 1) return SPAIR_ERR_UNSUPPORTED;
    if (B <= 0 || HW <= 0 || I <= 0) return SPAIR_ERR_SHAPE;
    hipLaunchKernelGGL(k_render_fwd, dim3(render_num_blocks(B, I)), dim3(256), 0, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon,
                       reinterpret_cast<float4*>(aux), bce_partial, B, HW, I, P, ac);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int render_bwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
               const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int C, int I,
               int P, int ac, float obj_scale, float alpha_scale, int g_bf16, hipStream_t s) {
    if (C != 1) return SPAIR_ERR_UNSUPPORTED;
    if ((P * P * 2) % 4 || (ld_s & 3) || (ld_g & 3)) return SPAIR_ERR_ALIGN;
    const size_t lds = (size_t)RB_WAVES * P * P * 2 * 2 * sizeof(float);
    const int kgroups = (HW + RB_WAVES - 1) / RB_WAVES;
    hipLaunchKernelGGL(k_render_bwd, dim3(B * kgroups), dim3(64 * RB_WAVES), lds, s, S, ld_s, nbox, pres, depth, ld_pd,
                       reinterpret_cast<const float4*>(aux), gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac, obj_scale,
                       alpha_scale, g_bf16);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

extern "C" int spair_render_fwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                                const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P,
                                int align_corners, void* stream) {
    return render_fwd(sprites, ld_s, nbox, pres, depth, 1, x, recon, aux, bce_partial, B, HW, C, I, P, align_corners,
                      (hipStream_t)stream);
}
extern "C" int spair_render_bwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                                const float* aux, const float* grad_loss, float* dlogits, float* dnbox, float* dpres,
                                float* ddepth, int B, int HW, int C, int I, int P, int align_corners, float obj_scale,
                                float alpha_scale, void* stream) {
    return render_bwd(sprites, ld_s, nbox, pres, depth, 1, aux, grad_loss, dlogits, dnbox, dpres, ddepth, ld_s, B, HW, C, I, P,
                      align_corners, obj_scale, alpha_scale, 0, (hipStream_t)stream);
}
