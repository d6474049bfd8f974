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
#include <stdlib.h>
#include "render_common.h"

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

template <bool S16>
__global__ __launch_bounds__(256) void k_render_fwd(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                    const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                    const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                    float* __restrict__ bce_partial, int B, int HW, int I, int P, int ac) {
    __shared__ Cand cand[RCH];
    __shared__ unsigned short wlist[4][RCH];     // per wave (= a 16 x 4 pixel strip of the tile): the candidates that reach its rows
    __shared__ int wave_cnt[4][5];               // [culling wave][tile, strip 0..3]
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
    const float bX = stn_base(min(px, I - 1), I, ac), bY = stn_base(min(py, I - 1), I, ac);   // this pixel's base coordinate, once
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
        // a second, finer cull per wave strip: at 16 x 16 a tile meets ~25 of 256 objects, a pixel ~9; the 4-row strips drop a
        // third of the (pixel, candidate) pairs.  Skipped pairs have zero weights, so the sums are unchanged to the bit.
        bool hs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sy0 = ty0 + 4 * q, sy1 = min(sy0 + 3, I - 1);
            hs[q] = hit && sy0 < I && src_of(c.ay, c.by, sy1, I, P, ac) > -1.f && src_of(c.ay, c.by, sy0, I, P, ac) < (float)P;
        }
        const unsigned long long bal = __ballot(hit);
        unsigned long long bs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bs[q] = __ballot(hs[q]);
        if (lane == 0) {
            wave_cnt[wave][0] = __popcll(bal);
#pragma unroll
            for (int q = 0; q < 4; ++q) wave_cnt[wave][1 + q] = __popcll(bs[q]);
        }
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wave_cnt[w][0];
        const unsigned long long below = (1ull << lane) - 1ull;
        const int ci_me = base + __popcll(bal & below);
        if (hit) cand[ci_me] = c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int bq = 0;
            for (int w = 0; w < wave; ++w) bq += wave_cnt[w][1 + q];
            if (hs[q]) wlist[q][bq + __popcll(bs[q] & below)] = (unsigned short)ci_me;
        }
        __syncthreads();
        const int nc = wave_cnt[0][1 + wave] + wave_cnt[1][1 + wave] + wave_cnt[2][1 + wave] + wave_cnt[3][1 + wave];
        const unsigned short* const wl = wlist[wave];
        // ---- accumulate the surviving objects at this thread's pixel.  Branch-free and software-pipelined: the four taps of
        // candidate ci+1 are in flight while candidate ci is composited (taps outside the sprite / pixels the object does not
        // cover read a clamped texel with weight 0).  With a `continue` per tap every load sat behind its own wait and the
        // kernel spent 80 % of its wave cycles parked (SQ_WAIT_ANY).
        if (inside && nc > 0) {
            float2 tv[4], tn[4];
            float tw[4], twn[4];
            float prs = 0.f, pdd = 0.f, prs_n = 0.f, pdd_n = 0.f;
            auto fetch = [&](int ci, float2 (&v)[4], float (&w)[4], float& pr_, float& pd_) {
                const Cand q = cand[ci];
                float gdum;
                const float sx = src_from_base(q.ax, q.bx, bX, P, ac, gdum), sy = src_from_base(q.ay, q.by, bY, P, ac, gdum);
                const bool cov = sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P;
                const float fx = floorf(sx), fy = floorf(sy);
                const int x0 = (int)fminf(fmaxf(fx, -1.f), (float)(P - 1)), y0 = (int)fminf(fmaxf(fy, -1.f), (float)(P - 1));
                const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const size_t sp = ((size_t)q.row * ld_s) >> 1;          // texel index of the sprite's first (grey, alpha) pair
                pr_ = q.pres; pd_ = q.pres * q.depth;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                    const bool ok = cov && yy >= 0 && yy < P && xx >= 0 && xx < P;
                    w[t] = ok ? ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0) : 0.f;
                    v[t] = ld_texel<S16>(S, sp + (min(max(yy, 0), P - 1) * P + min(max(xx, 0), P - 1)));
                }
            };
            fetch(wl[0], tv, tw, prs, pdd);
            for (int ci = 0; ci < nc; ++ci) {
                fetch(wl[min(ci + 1, nc - 1)], tn, twn, prs_n, pdd_n);
                float g = 0.f, a = 0.f, m = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    g += tw[t] * tv[t].x;
                    a += tw[t] * (tv[t].y * prs);
                    m += tw[t] * fmaxf(tv[t].y * pdd, 0.01f);
                }
                num += g * a * (m + 1e-9f);
                den += m;
#pragma unroll
                for (int t = 0; t < 4; ++t) { tv[t] = tn[t]; tw[t] = twn[t]; }
                prs = prs_n; pdd = pdd_n;
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
            aux[pi] = make_float2(gr * invD, pre);        // (dBCE/dpre / D, pre)
        }
    }
    bce = block_reduce_sum_256(bce, red);
    if (threadIdx.x == 0) bce_partial[blockIdx.x] = bce;
}

// ---------------------------------------------------------------------------------------------
// backward: one WORKGROUP (4 waves) per object, no atomics (LDS float atomics retire ~1 lane/clk on gfx950 and
// were 60% of the first scatter kernel).  Two passes per object, both conflict-free:
//   A  threads = pixels of the object's footprint: re-sample the sprite from a zero-bordered LDS copy that already
//      holds (grey, alpha, importance) per texel -- no tap bounds tests -- form the three per-pixel adjoints,
//      reduce d(z_where) in registers, park the adjoints in LDS.  The per-pixel aux record is prefetched one
//      iteration ahead (the only global load in the loop);
//   B  threads = sprite texels: each texel GATHERS the pixels whose bilinear hat covers it (the transpose of the
//      4-tap scatter), accumulates d(pres, depth), applies sigmoid' and the logit scales and writes dlogits once.
// The staged footprint is RB_CAP pixels; larger footprints are processed per texel tile (pixels on tile seams are
// re-sampled; their z_where gradients are counted by the owning tile only), and a tile that pass B covers in one
// sweep whose footprint still does not fit is streamed in row chunks with the texel sums held in registers.
// LDS per workgroup is 26.7 KB at P=28, so 6 workgroups = 24 waves share a CU.
// Workgroup -> sample mapping is XCD-aware: every object of sample b runs on XCD b%8, so the per-pixel aux map of a
// sample (I*I*16 B) is only ever cached in one L2.
// ---------------------------------------------------------------------------------------------
#define RB_WAVES 4
#define RB_CAP 1024
#define RB_T (64 * RB_WAVES)
#define RB_TSH (RB_WAVES == 1 ? 6 : RB_WAVES == 2 ? 7 : RB_WAVES == 4 ? 8 : 9)
__device__ __forceinline__ int rb_span(int T, float inv, int I) { return min((int)ceilf((float)(T + 1) * inv) + 3, I); }
__device__ __forceinline__ int rb_lane_shift(int w) { return w <= 16 ? 4 : (w <= 32 ? 5 : 6); }
// sweeps pass B needs for a TU x TV texel tile (a workgroup covers (RB_T/CW) rows x CW columns per sweep)
__device__ __forceinline__ int rb_sweeps(int TU, int TV) {
    const int sh = rb_lane_shift(TU), cw = 1 << sh, rsh = RB_TSH - sh, rows = 1 << rsh;
    return ((TU + cw - 1) >> sh) * ((TV + rows - 1) >> rsh);
}

template <bool S16>
__global__ __launch_bounds__(RB_T) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_render_bwd(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                     const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                     const float2* __restrict__ aux, const float* __restrict__ gloss,
                                                     float* __restrict__ dlogits, float* __restrict__ dnbox, float* __restrict__ dpres,
                                                     float* __restrict__ ddepth, int ld_g, int B, int HW, int I, int P, int ac,
                                                     float obj_scale, float alpha_scale, int g_bf16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[RB_WAVES][6];
    __shared__ float geo[16];
    const int PS = P + 2, NTX = PS * PS;                     // zero-bordered sprite
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float4* Ssh = reinterpret_cast<float4*>(sm);             // (grey, alpha, importance, -)
    float* pb = reinterpret_cast<float*>(Ssh + NTX);          // staged per-pixel adjoints [RB_CAP][3]
    float* btab = pb + 3 * RB_CAP;                            // base coordinate of output index j (no division per pixel)
    for (int e = tid; e < I; e += RB_T) btab[e] = stn_base(e, I, ac);
    // (sample, object) of this workgroup: grid = (B, HW); consecutive workgroup ids walk the samples, so with B % 8 == 0
    // every object of sample b lands on XCD b % 8
    const int b = blockIdx.x, k = blockIdx.y;
    const int r = k * B + b;
    const float pr = pres[(size_t)r * ld_pd], dp = depth[(size_t)r * ld_pd], pd = pr * dp;
    const float gl = *gloss;
    const float2* auxb = aux + (size_t)b * I * I;
    const float mult = ac ? 0.5f * (float)(P - 1) : 0.5f * (float)P;
    // The object's geometry is the same for every thread, and gfx9 has no scalar float ALU: computed by all four waves it is ~300
    // VALU instructions per wave and object in a kernel that is VALU-issue bound.  Wave 0 computes it while the others already
    // fetch the sprite; everybody reads it from LDS behind the staging barrier.
    if (wave == 0) {
        const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
        const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
        const float ax_ = 1.f / nb.z, bx_ = -tx / nb.z, ay_ = 1.f / nb.w, by_ = -ty / nb.w;
        float sx0_, sxa_, sy0_, sya_;
        src_affine(ax_, bx_, I, P, ac, sx0_, sxa_);
        src_affine(ay_, by_, I, P, ac, sy0_, sya_);
        const float isx_ = __builtin_amdgcn_rcpf(sxa_), isy_ = __builtin_amdgcn_rcpf(sya_);   // only used for (conservative) index bounds
        int TU_ = P, TV_ = P;                                     // texel tile whose pixel footprint fits the staging buffer
        while (rb_span(TU_, isx_, I) * rb_span(TV_, isy_, I) > RB_CAP && rb_sweeps(TU_, TV_) > 1) {
            if (rb_span(TV_, isy_, I) >= rb_span(TU_, isx_, I) && TV_ > 1) TV_ = (TV_ + 1) >> 1;
            else if (TU_ > 1) TU_ = (TU_ + 1) >> 1;
            else TV_ = (TV_ + 1) >> 1;
        }
        if (lane == 0) {
            geo[0] = ax_; geo[1] = bx_; geo[2] = ay_; geo[3] = by_; geo[4] = sx0_; geo[5] = sxa_; geo[6] = sy0_; geo[7] = sya_;
            geo[8] = isx_; geo[9] = isy_; geo[10] = -mult * ax_; geo[11] = -mult * ay_;      // d(source coord)/d(t) incl. the unnormalisation
            geo[12] = __int_as_float(TU_); geo[13] = __int_as_float(TV_);
        }
    }
    if (P <= 32) {
        // all loads are issued before the first LDS write
        constexpr int RV = RB_T / 32;                         // rows per sweep
        constexpr int NI = 32 / RV;
        const size_t Sr = ((size_t)r * ld_s) >> 1;
        const int u = tid & 31;
        float2 t[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = RV * i + (tid >> 5);
            t[i] = ld_texel<S16>(S, Sr + min(v, P - 1) * P + min(u, P - 1));
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = RV * i + (tid >> 5);
            if (v < P && u < P) Ssh[(v + 1) * PS + u + 1] = make_float4(t[i].x, t[i].y, fmaxf(t[i].y * pd, 0.01f), 0.f);
        }
    } else {
        const size_t Sr = ((size_t)r * ld_s) >> 1;
        for (int e = tid; e < P * P; e += RB_T) {
            const float2 t = ld_texel<S16>(S, Sr + e);
            const int v = e / P, u = e - v * P;
            Ssh[(v + 1) * PS + u + 1] = make_float4(t.x, t.y, fmaxf(t.y * pd, 0.01f), 0.f);
        }
    }
    for (int e = tid; e < PS; e += RB_T) {
        Ssh[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        Ssh[(PS - 1) * PS + e] = make_float4(0.f, 0.f, 0.f, 0.f);
        Ssh[e * PS] = make_float4(0.f, 0.f, 0.f, 0.f);
        Ssh[e * PS + PS - 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float g_tx = 0.f, g_ty = 0.f, g_xs = 0.f, g_ys = 0.f, g_pr = 0.f, g_dp = 0.f;
    // source coordinate of pixel (px,py) ~ (sx0 + px*sxa, sy0 + py*sya), slopes > 0: used for index BOUNDS only (each
    // bound keeps a spare pixel); the coordinates themselves come from src_from_base like the forward's
    const float ax = geo[0], bx = geo[1], ay = geo[2], by = geo[3], sx0 = geo[4], sy0 = geo[6];
    const float isx = geo[8], isy = geo[9], cgx = geo[10], cgy = geo[11];
    const int TU = __float_as_int(geo[12]), TV = __float_as_int(geo[13]);
    for (int tv0 = 0; tv0 < P; tv0 += TV)
    for (int tu0 = 0; tu0 < P; tu0 += TU) {
        const int tu1 = min(tu0 + TU, P), tv1 = min(tv0 + TV, P);      // texel tile [tu0,tu1) x [tv0,tv1)
        // pixels whose source coordinate lies in (tu0-1, tu1) x (tv0-1, tv1), one spare pixel either side
        const int PX0 = max((int)floorf(((float)(tu0 - 1) - sx0) * isx), 0), PX1 = min((int)ceilf(((float)tu1 - sx0) * isx), I - 1);
        const int PY0 = max((int)floorf(((float)(tv0 - 1) - sy0) * isy), 0), PY1 = min((int)ceilf(((float)tv1 - sy0) * isy), I - 1);
        const int pw = PX1 - PX0 + 1, ph = PY1 - PY0 + 1;
        const bool empty = pw <= 0 || ph <= 0;
        const int rows_per = empty ? 1 : max(1, RB_CAP / pw);           // host guarantees I <= RB_CAP
        const bool multi = !empty && ph > rows_per;                      // only reachable when pass B is a single sweep
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const int wsh = rb_lane_shift(pw), W = 1 << wsh, RPI = 64 >> wsh;
        const int lx = lane & (W - 1), ly = lane >> wsh;
        const int nxc = empty ? 1 : (pw + W - 1) >> wsh;
        const int csh = rb_lane_shift(tu1 - tu0), CW = 1 << csh, CR = RB_T >> csh;
        const int cu = tid & (CW - 1), cv = tid >> csh;
        // z_where gradients are counted by the tile that owns the pixel: source coord in [tu0, tu1) x [tv0, tv1),
        // the first / last tile also taking the half-covered border pixels
        const float oxl = tu0 == 0 ? -1.f : (float)tu0, oxh = (float)tu1, oyl = tv0 == 0 ? -1.f : (float)tv0, oyh = (float)tv1;
        for (int cy0 = PY0; cy0 <= (empty ? PY0 : PY1); cy0 += rows_per) {
            const int cy1 = empty ? cy0 - 1 : min(cy0 + rows_per - 1, PY1);
            // ---- pass A: a wave covers (64/W) rows x W columns per iteration; iterations are dealt round-robin to waves
            const int nj = ((cy1 - cy0 + RPI) >> (6 - wsh)) * nxc;
            auto pixel_of = [&](int j, int& px, int& py) {
                const int yi = nxc == 1 ? j : j / nxc, xi = j - yi * nxc;
                px = PX0 + (xi << wsh) + lx;
                py = cy0 + yi * RPI + ly;
            };
            float2 avn = make_float2(0.f, 0.f);
            if (wave < nj) {
                int px, py;
                pixel_of(wave, px, py);
                avn = auxb[min(py, cy1) * I + min(px, PX1)];
            }
            for (int j = wave; j < nj; j += RB_WAVES) {
                const float2 av = avn;                                   // (dBCE/dpre / D, pre)
                int px, py;
                {
                    pixel_of(min(j + RB_WAVES, nj - 1), px, py);
                    avn = auxb[min(py, cy1) * I + min(px, PX1)];
                }
                pixel_of(j, px, py);
                if (py > cy1 || px > PX1) continue;
                float gxn, gyn;
                const float sx = src_from_base(ax, bx, btab[px], P, ac, gxn), sy = src_from_base(ay, by, btab[py], P, ac, gyn);
                const bool inside = sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P;
                const float fx = fminf(fmaxf(floorf(sx), -1.f), (float)(P - 1)), fy = fminf(fmaxf(floorf(sy), -1.f), (float)(P - 1));
                const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const float4* t = Ssh + ((int)fy + 1) * PS + (int)fx + 1;
                const float4 t0 = t[0], t1 = t[1], t2 = t[PS], t3 = t[PS + 1];
                // separable bilinear: x first, then y; the y-derivative is (bottom - top)
                const float gT = t0.x * wx0 + t1.x * wx1, gB = t2.x * wx0 + t3.x * wx1;
                const float aT = t0.y * wx0 + t1.y * wx1, aB = t2.y * wx0 + t3.y * wx1;
                const float mT = t0.z * wx0 + t1.z * wx1, mB = t2.z * wx0 + t3.z * wx1;
                const float g = gT * wy0 + gB * wy1, a = (aT * wy0 + aB * wy1) * pr, m = mT * wy0 + mB * wy1;
                const float go = inside ? av.x * gl : 0.f;              // dBCE/dpre / D
                const float Dm = m + 1e-9f;
                const float d_g = go * a * Dm;
                const float d_a = go * g * Dm;                           // wrt (alpha*pres)
                const float d_m = go * (a * g - av.y);
                float* q = pb + ((py - cy0) * pw + (px - PX0)) * 3;
                q[0] = d_g; q[1] = d_a; q[2] = d_m;
                if (!(sx >= oxl && sx < oxh && sy >= oyl && sy < oyh)) continue;
                const float dap = d_a * pr;
                const float dgx = (t1.x - t0.x) * wy0 + (t3.x - t2.x) * wy1, dax = (t1.y - t0.y) * wy0 + (t3.y - t2.y) * wy1,
                            dmx = (t1.z - t0.z) * wy0 + (t3.z - t2.z) * wy1;
                const float g_sx = d_g * dgx + dap * dax + d_m * dmx;               // d/d(source x), pixel units
                const float g_sy = d_g * (gB - gT) + dap * (aB - aT) + d_m * (mB - mT);
                g_tx += g_sx; g_xs = fmaf(g_sx, gxn, g_xs);              // scaled by cgx / cgy after the loop
                g_ty += g_sy; g_ys = fmaf(g_sy, gyn, g_ys);
            }
            __syncthreads();
            // ---- pass B: threads = texels of the tile, (RB_T/CW) rows x CW columns; gather the pixels under each hat
            const bool last = empty || cy1 == PY1;
            for (int ub = tu0; ub < tu1; ub += CW) {
                const int u = ub + cu;
                const int xa = max(PX0, (int)floorf(((float)(u - 1) - sx0) * isx) + 1), xe = min(PX1, (int)ceilf(((float)(u + 1) - sx0) * isx) - 1);
                for (int vb = tv0; vb < tv1; vb += CR) {
                    const int v = vb + cv;
                    if (u >= tu1 || v >= tv1) continue;
                    const int ya = max(cy0, (int)floorf(((float)(v - 1) - sy0) * isy) + 1), ye = min(cy1, (int)ceilf(((float)(v + 1) - sy0) * isy) - 1);
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                    for (int py = ya; py <= ye; ++py) {
                        float gdum;
                        const float wy = fmaxf(1.f - fabsf(src_from_base(ay, by, btab[py], P, ac, gdum) - (float)v), 0.f);
                        const float* q = pb + ((py - cy0) * pw - PX0) * 3;
                        for (int px = xa; px <= xe; ++px) {
                            const float w = wy * fmaxf(1.f - fabsf(src_from_base(ax, bx, btab[px], P, ac, gdum) - (float)u), 0.f);
                            s0 = fmaf(w, q[px * 3], s0);
                            s1 = fmaf(w, q[px * 3 + 1], s1);
                            s2 = fmaf(w, q[px * 3 + 2], s2);
                        }
                    }
                    if (multi) { a0 += s0; a1 += s1; a2 += s2; s0 = a0; s1 = a1; s2 = a2; }
                    if (!last) continue;
                    const float4 sv = Ssh[(v + 1) * PS + u + 1];
                    const bool act = (sv.y * pd) >= 0.01f;                   // importance not clamped
                    const float s2a = act ? s2 * sv.y : 0.f;
                    g_pr += s1 * sv.y + s2a * dp;
                    g_dp += s2a * pr;
                    // through the analytical sigmoid and the logit scales (models.py:485-492)
                    const int e = (v * P + u) * 2;
                    const float ox = s0 * sv.x * (1.f - sv.x) * obj_scale;
                    const float oy = (s1 * pr + (act ? s2 * pd : 0.f)) * sv.y * (1.f - sv.y) * alpha_scale;
                    if (g_bf16) {
                        __bf16 ob[2] = {(__bf16)ox, (__bf16)oy};
                        *reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(dlogits) + (size_t)r * ld_g + e) = *reinterpret_cast<unsigned*>(ob);
                    } else {
                        *reinterpret_cast<float2*>(dlogits + (size_t)r * ld_g + e) = make_float2(ox, oy);
                    }
                }
            }
            __syncthreads();
        }
    }
    g_tx = wave_reduce_sum(g_tx) * cgx; g_ty = wave_reduce_sum(g_ty) * cgy;
    g_xs = wave_reduce_sum(g_xs) * cgx; g_ys = wave_reduce_sum(g_ys) * cgy;
    g_pr = wave_reduce_sum(g_pr); g_dp = wave_reduce_sum(g_dp);
    if (lane == 0) {
        red[wave][0] = g_tx; red[wave][1] = g_ty; red[wave][2] = g_xs; red[wave][3] = g_ys; red[wave][4] = g_pr; red[wave][5] = g_dp;
    }
    __syncthreads();
    if (tid == 0) {
        float o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            o[i] = 0.f;
            for (int w = 0; w < RB_WAVES; ++w) o[i] += red[w][i];
        }
        *reinterpret_cast<float4*>(dnbox + (size_t)r * 4) = make_float4(2.f * o[0], 2.f * o[1], o[2], o[3]);
        dpres[r] = o[4];
        ddepth[r] = o[5];
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

int render_fwd2(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x,
                float* recon, float* aux, float* bce_partial, int B, int HW, int I, int P, int ac, int s_bf16, hipStream_t s);
int render_bwd2(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
                const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int I, int P,
                int ac, float obj_scale, float alpha_scale, const void* rec, hipStream_t s);
int render_prep(const float* nbox, const float* pres, const float* depth, int ld_pd, void* rec, int B, int HW, int I, int P, int ac,
                hipStream_t s);
int render_fwd_mma(const void* S16, int ld_s, const void* rec, const float* x, float* recon, float* aux, float* bce_partial, int B, int HW,
                   int I, int P, int ac, hipStream_t s);
// s_bf16: sprites are bf16 (grey, alpha) pairs; ld_s stays in ELEMENTS of that type.  aux: B*I*I float2 (dBCE/dpre / D, pre).
int render_fwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x,
               float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int ac, int s_bf16, hipStream_t s) {
    if (C != 1) return SPAIR_ERR_UNSUPPORTED;
    if (B <= 0 || HW <= 0 || I <= 0 || (ld_s & 1)) return SPAIR_ERR_SHAPE;
    {
        const int rc = render_fwd2(S, ld_s, nbox, pres, depth, ld_pd, x, recon, aux, bce_partial, B, HW, I, P, ac, s_bf16, s);
        if (rc != SPAIR_ERR_UNSUPPORTED) return rc;
    }
    if (s_bf16)
        hipLaunchKernelGGL(k_render_fwd<true>, dim3(render_num_blocks(B, I)), dim3(256), 0, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon,
                           reinterpret_cast<float2*>(aux), bce_partial, B, HW, I, P, ac);
    else
    hipLaunchKernelGGL(k_render_fwd<false>, dim3(render_num_blocks(B, I)), dim3(256), 0, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon,
                       reinterpret_cast<float2*>(aux), bce_partial, B, HW, I, P, ac);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int render_bwd(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
               const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int C, int I,
               int P, int ac, float obj_scale, float alpha_scale, int g_bf16, int s_bf16, const void* rec, hipStream_t s) {
    if (C != 1) return SPAIR_ERR_UNSUPPORTED;
    if ((ld_s & 1) || (ld_g & 1)) return SPAIR_ERR_ALIGN;
    if (g_bf16 && s_bf16) {   // the bf16 step: one wave per object, sampling on VALU, its transpose on the matrix cores
        // rec (optional): the per-object records the forward's k_render_prep left in the workspace
        const int rc = render_bwd2(S, ld_s, nbox, pres, depth, ld_pd, aux, gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac,
                                   obj_scale, alpha_scale, rec, s);
        if (rc != SPAIR_ERR_UNSUPPORTED) return rc;
    }
    if (I > RB_CAP || (long long)I * I > 0x7fffffffLL / 4 || HW > 65535) return SPAIR_ERR_UNSUPPORTED;
    const size_t lds = ((size_t)(P + 2) * (P + 2) * 4 + 3 * RB_CAP + I) * sizeof(float);
    if (lds > 65536) return SPAIR_ERR_UNSUPPORTED;
    if (s_bf16)
        hipLaunchKernelGGL(k_render_bwd<true>, dim3(B, HW), dim3(RB_T), lds, s, S, ld_s, nbox, pres, depth, ld_pd,
                           reinterpret_cast<const float2*>(aux), gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac, obj_scale,
                           alpha_scale, g_bf16);
    else
    hipLaunchKernelGGL(k_render_bwd<false>, dim3(B, HW), dim3(RB_T), lds, s, S, ld_s, nbox, pres, depth, ld_pd,
                       reinterpret_cast<const float2*>(aux), gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac, obj_scale,
                       alpha_scale, g_bf16);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

extern "C" int spair_render_fwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                                const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P,
                                int align_corners, void* stream) {
    return render_fwd(sprites, ld_s, nbox, pres, depth, 1, x, recon, aux, bce_partial, B, HW, C, I, P, align_corners, 0,
                      (hipStream_t)stream);
}
extern "C" int spair_render_bwd(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth,
                                const float* aux, const float* grad_loss, float* dlogits, float* dnbox, float* dpres,
                                float* ddepth, int B, int HW, int C, int I, int P, int align_corners, float obj_scale,
                                float alpha_scale, void* stream) {
    return render_bwd(sprites, ld_s, nbox, pres, depth, 1, aux, grad_loss, dlogits, dnbox, dpres, ddepth, ld_s, B, HW, C, I, P,
                      align_corners, obj_scale, alpha_scale, 0, 0, nullptr, (hipStream_t)stream);
}

// 16-bit sprite variants (what the bf16 training step runs): sprites are fp16 (grey, alpha) pairs [N][ld_s] (ld_s in elements), the
// d-logits come back as bf16 [N][ld_s]
extern "C" int spair_render_fwd16(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth,
                                  const float* x, float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P,
                                  int align_corners, void* stream) {
    return render_fwd(reinterpret_cast<const float*>(sprites_f16), ld_s, nbox, pres, depth, 1, x, recon, aux, bce_partial, B, HW, C, I, P,
                      align_corners, 1, (hipStream_t)stream);
}
extern "C" int spair_render_bwd16(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth,
                                  const float* aux, const float* grad_loss, void* dlogits_bf16, float* dnbox, float* dpres,
                                  float* ddepth, int B, int HW, int C, int I, int P, int align_corners, float obj_scale,
                                  float alpha_scale, void* stream) {
    return render_bwd(reinterpret_cast<const float*>(sprites_f16), ld_s, nbox, pres, depth, 1, aux, grad_loss,
                      reinterpret_cast<float*>(dlogits_bf16), dnbox, dpres, ddepth, ld_s, B, HW, C, I, P, align_corners, obj_scale,
                      alpha_scale, 1, 1, nullptr, (hipStream_t)stream);
}
// The matrix-core forward renderer of the bf16 step (render3.hip): spair_render_prep writes the per-object records (32 bytes each,
// B * HW of them, caller-owned), spair_render_fwd16m composites from them.  SPAIR_ERR_UNSUPPORTED (P != 28, align_corners, HW > 1024):
// use spair_render_fwd16.
extern "C" int spair_render_prep(const float* nbox, const float* pres, const float* depth, void* records, int B, int HW, int I, int P,
                                 int align_corners, void* stream) {
    if (B <= 0 || HW <= 0 || I <= 0) return SPAIR_ERR_SHAPE;
    return render_prep(nbox, pres, depth, 1, records, B, HW, I, P, align_corners, (hipStream_t)stream);
}
extern "C" int spair_render_fwd16m(const void* sprites_f16, int ld_s, const void* records, const float* x, float* recon, float* aux,
                                   float* bce_partial, int B, int HW, int C, int I, int P, int align_corners, void* stream) {
    if (C != 1) return SPAIR_ERR_UNSUPPORTED;
    if (B <= 0 || HW <= 0 || I <= 0) return SPAIR_ERR_SHAPE;
    return render_fwd_mma(sprites_f16, ld_s, records, x, recon, aux, bce_partial, B, HW, I, P, align_corners, (hipStream_t)stream);
}
// spair_render_bwd16 reading the inverse-affine parameters and footprints from the records spair_render_prep wrote for the same nbox /
// pres / depth (what the training step does); same outputs.
extern "C" int spair_render_bwd16r(const void* sprites_f16, int ld_s, const float* nbox, const float* pres, const float* depth,
                                   const void* records, const float* aux, const float* grad_loss, void* dlogits_bf16, float* dnbox,
                                   float* dpres, float* ddepth, int B, int HW, int C, int I, int P, int align_corners, float obj_scale,
                                   float alpha_scale, void* stream) {
    return render_bwd(reinterpret_cast<const float*>(sprites_f16), ld_s, nbox, pres, depth, 1, aux, grad_loss,
                      reinterpret_cast<float*>(dlogits_bf16), dnbox, dpres, ddepth, ld_s, B, HW, C, I, P, align_corners, obj_scale,
                      alpha_scale, 1, 1, records, (hipStream_t)stream);
}
