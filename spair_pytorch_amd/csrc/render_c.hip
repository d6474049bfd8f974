// K6 for images with more than one colour channel (reference: models.py:452-547 with cfg.INPUT_IMAGE_SHAPE[0] = C > 1; stn(inverse=True)
// modules.py:239-269).  The three tuned renderer generations (render.hip, render2.hip, render3.hip) are written for the reference's
// greyscale data: (grey, alpha) texel pairs.  These two kernels are the same mathematics with CH = C + 1 channels per texel -- sprites
// [N][P*P][CH] fp32 = (colour_0 .. colour_{C-1}, alpha) after the sigmoid, images / reconstructions [B][C][I][I] -- written for
// correctness and run-to-run determinism, not speed (no tuned RGB path exists in this round; the BASELINE configs are greyscale):
//   * forward, pixel-centric as render.hip: one workgroup per (sample, 16 x 16 tile) culls the objects whose zero-padded footprint meets
//     the tile into LDS, every thread composites its pixel:  pre_c = sum_k g_kc a_k (m_k + 1e-9) / (sum_k m_k + HW 1e-9),
//     g = warped colour, a = warped alpha * pres, m = warped max(alpha * pres * depth, 0.01)  (taps on the padding are zero);
//     per (pixel, channel): reconstruction, BCE, and the record (dBCE/dpre / D, pre) for the backward;
//   * backward, object-centric: ONE WAVE per object walks the object's pixel footprint, forms the per-pixel adjoints of (g_c, a, m),
//     accumulates d z_where in registers and scatters the tap adjoints into an LDS image [P*P][C + 2] with ds_add_f32 -- one wave, so
//     the order of the additions is the program's: bit-identical from run to run; then per texel sigmoid', the logit scales, d pres /
//     d depth and ONE store of the d-logits.
#include "render_common.h"

namespace {

constexpr int RC_MAXC = 3;

template <int C>
__global__ __launch_bounds__(256) void k_render_fwd_c(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                      const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                      const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                      float* __restrict__ bce_partial, int B, int HW, int I, int P, int ac) {
    constexpr int CH = C + 1;
    __shared__ Cand cand[RCH];
    __shared__ float red[4];
    const int tiles_x = (I + RT - 1) / RT, tiles = tiles_x * tiles_x;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tx0 = (tile % tiles_x) * RT, ty0 = (tile / tiles_x) * RT;
    const int lx = threadIdx.x & (RT - 1), ly = threadIdx.x >> 4;
    const int px = tx0 + lx, py = ty0 + ly;
    const bool inside = px < I && py < I;
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
    float num[C], den = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) num[c] = 0.f;
    const float bX = stn_base(min(px, I - 1), I, ac), bY = stn_base(min(py, I - 1), I, ac);
    for (int k0 = 0; k0 < HW; k0 += RCH) {
        const int k = k0 + threadIdx.x;
        if (k < HW) {
            const int r = k * B + b;
            const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
            const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
            Cand c;
            c.ax = 1.f / nb.z; c.bx = -tx / nb.z; c.ay = 1.f / nb.w; c.by = -ty / nb.w;
            c.pres = pres[(size_t)r * ld_pd]; c.depth = depth[(size_t)r * ld_pd]; c.row = r;
            const bool hit = src_of(c.ax, c.bx, tx1, I, P, ac) > -1.f && src_of(c.ax, c.bx, tx0, I, P, ac) < (float)P &&
                             src_of(c.ay, c.by, ty1, I, P, ac) > -1.f && src_of(c.ay, c.by, ty0, I, P, ac) < (float)P;
            if (!hit) c.row = -1;                     // (every thread its own slot: the order of the sums is the object order)
            cand[threadIdx.x] = c;
        } else {
            cand[threadIdx.x].row = -1;
        }
        __syncthreads();
        if (inside) {
            for (int ci = 0; ci < RCH && k0 + ci < HW; ++ci) {
                const Cand q = cand[ci];
                if (q.row < 0) continue;                                  // (workgroup-uniform)
                float gdum;
                const float sx = src_from_base(q.ax, q.bx, bX, P, ac, gdum), sy = src_from_base(q.ay, q.by, bY, P, ac, gdum);
                if (!(sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P)) continue;
                const float fx = floorf(sx), fy = floorf(sy);
                const int x0 = (int)fx, y0 = (int)fy;
                const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
                const float* sp = S + (size_t)q.row * ld_s;
                const float pd = q.pres * q.depth;
                float g[C], a = 0.f, m = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) g[c] = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                    if (yy < 0 || yy >= P || xx < 0 || xx >= P) continue;
                    const float w = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                    const float* tp = sp + (size_t)(yy * P + xx) * CH;
#pragma unroll
                    for (int c = 0; c < C; ++c) g[c] += w * tp[c];
                    const float al = tp[C];
                    a += w * (al * q.pres);
                    m += w * fmaxf(al * pd, 0.01f);
                }
#pragma unroll
                for (int c = 0; c < C; ++c) num[c] += g[c] * a * (m + 1e-9f);
                den += m;
            }
        }
        __syncthreads();
    }
    float bce = 0.f;
    if (inside) {
        const float D = den + (float)HW * 1e-9f;          // every object adds 1e-9 (models.py:527)
        const float invD = 1.f / D;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float pre = num[c] * invD;
            const float r = fminf(fmaxf(pre, 0.f), 1.f);
            const size_t pi = (((size_t)b * C + c) * I + py) * I + px;
            const float xv = x[pi];
            recon[pi] = r;
            bce += -(xv * fmaxf(logf(r), -100.f) + (1.f - xv) * fmaxf(logf(1.f - r), -100.f));      // torch BCE: log clamped at -100
            if (aux) {
                const float gr = (pre >= 0.f && pre <= 1.f) ? (r - xv) / fmaxf(r * (1.f - r), 1e-12f) : 0.f;
                aux[pi] = make_float2(gr * invD, pre);
            }
        }
    }
    bce = block_reduce_sum_256(bce, red);
    if (threadIdx.x == 0) bce_partial[blockIdx.x] = bce;
}

template <int C>
__global__ __launch_bounds__(64) void k_render_bwd_c(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                     const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                     const float2* __restrict__ aux, const float* __restrict__ gloss,
                                                     float* __restrict__ dlogits, float* __restrict__ dnbox, float* __restrict__ dpres,
                                                     float* __restrict__ ddepth, int ld_g, int B, int HW, int I, int P, int ac,
                                                     float obj_scale, float alpha_scale) {
    constexpr int CH = C + 1, NA = C + 2;        // texel channels; adjoint channels (colour.., alpha * pres, importance)
    extern __shared__ float acc[];               // [P * P][NA]
    const int lane = threadIdx.x;
    const int b = blockIdx.x, k = blockIdx.y, r = k * B + b;
    for (int e = lane; e < P * P * NA; e += 64) acc[e] = 0.f;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    const float ax = 1.f / nb.z, bx = -tx / nb.z, ay = 1.f / nb.w, by = -ty / nb.w;
    const float pr = pres[(size_t)r * ld_pd], dp = depth[(size_t)r * ld_pd], pd = pr * dp;
    const float gl = *gloss;
    const float mult = ac ? 0.5f * (float)(P - 1) : 0.5f * (float)P;
    // pixel footprint: the indices whose source coordinate lies in (-1, P), exact w.r.t. the forward's own coordinate formula
    float sx0, sxa, sy0, sya;
    src_affine(ax, bx, I, P, ac, sx0, sxa);
    src_affine(ay, by, I, P, ac, sy0, sya);
    int PX0, PX1, PY0, PY1;
    const float inv_I = 1.f / (float)I;
    if (ac) { rb2_range<1, 0>(ax, bx, sx0, 1.f / sxa, I, inv_I, P, PX0, PX1); rb2_range<1, 0>(ay, by, sy0, 1.f / sya, I, inv_I, P, PY0, PY1); }
    else { rb2_range<0, 0>(ax, bx, sx0, 1.f / sxa, I, inv_I, P, PX0, PX1); rb2_range<0, 0>(ay, by, sy0, 1.f / sya, I, inv_I, P, PY0, PY1); }
    const int pw = max(PX1 - PX0 + 1, 0), ph = max(PY1 - PY0 + 1, 0);
    const float* sp = S + (size_t)r * ld_s;
    __builtin_amdgcn_s_waitcnt(0xc07f);      // this lane's zeroing stores
    __builtin_amdgcn_wave_barrier();
    float g_tx = 0.f, g_ty = 0.f, g_xs = 0.f, g_ys = 0.f;
    for (int i0 = 0; i0 < pw * ph; i0 += 64) {
        const int idx = i0 + lane;
        const bool live = idx < pw * ph;
        const int iy = min(idx, pw * ph - 1) / pw, ix = min(idx, pw * ph - 1) - iy * pw;
        const int px = PX0 + ix, py = PY0 + iy;
        float gnx, gny;
        const float sx = src_from_base(ax, bx, stn_base(px, I, ac), P, ac, gnx), sy = src_from_base(ay, by, stn_base(py, I, ac), P, ac, gny);
        const bool cov = live && sx > -1.f && sx < (float)P && sy > -1.f && sy < (float)P;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = sx - fx, wy1 = sy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        float g[C], dgx[C], dgy[C], a = 0.f, dax = 0.f, day = 0.f, m = 0.f, dmx = 0.f, dmy = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { g[c] = 0.f; dgx[c] = 0.f; dgy[c] = 0.f; }
        float wt[4];
        int tt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
            const bool ok = cov && yy >= 0 && yy < P && xx >= 0 && xx < P;
            const float wy = (t >> 1) ? wy1 : wy0, wx = (t & 1) ? wx1 : wx0;
            wt[t] = ok ? wy * wx : 0.f;
            tt[t] = ok ? yy * P + xx : -1;
            if (!ok) continue;
            const float sgx = (t & 1) ? wy : -wy, sgy = (t >> 1) ? wx : -wx;      // d weight / d (source x, y)
            const float* tp = sp + (size_t)tt[t] * CH;
#pragma unroll
            for (int c = 0; c < C; ++c) { const float v = tp[c]; g[c] += wt[t] * v; dgx[c] += sgx * v; dgy[c] += sgy * v; }
            const float al = tp[C], ap = al * pr, mm = fmaxf(al * pd, 0.01f);      // (the forward's own operations, in its order)
            a += wt[t] * ap; dax += sgx * ap; day += sgy * ap;
            m += wt[t] * mm; dmx += sgx * mm; dmy += sgy * mm;
        }
        float d_g[C], d_a = 0.f, d_m = 0.f;
        {
            const size_t pix = (size_t)py * I + px;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float2 av = cov ? aux[((size_t)b * C + c) * I * I + pix] : make_float2(0.f, 0.f);
                const float go = av.x * gl;        // dBCE/dpre_c / D
                const float tq = go * (m + 1e-9f);
                d_g[c] = tq * a;
                d_a += tq * g[c];
                d_m += go * (a * g[c] - av.y);
            }
        }
        float g_sx = d_a * dax + d_m * dmx, g_sy = d_a * day + d_m * dmy;
#pragma unroll
        for (int c = 0; c < C; ++c) { g_sx += d_g[c] * dgx[c]; g_sy += d_g[c] * dgy[c]; }
        g_tx += g_sx; g_xs = fmaf(g_sx, gnx, g_xs);
        g_ty += g_sy; g_ys = fmaf(g_sy, gny, g_ys);
        // tap adjoints -> the LDS image (one wave: the additions happen in program order, lane by lane)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (tt[t] < 0) continue;
            float* q = acc + tt[t] * NA;
#pragma unroll
            for (int c = 0; c < C; ++c) atomicAdd(q + c, wt[t] * d_g[c]);
            atomicAdd(q + C, wt[t] * d_a);
            atomicAdd(q + C + 1, wt[t] * d_m);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // per texel: through the analytical sigmoid and the logit scales (models.py:485-492); d pres, d depth
    float g_pr = 0.f, g_dp = 0.f;
    for (int e = lane; e < P * P; e += 64) {
        const float* tp = sp + (size_t)e * CH;
        const float* q = acc + e * NA;
        const float sa = tp[C], s1 = q[C], s2 = q[C + 1];
        const bool act = (sa * pd) >= 0.01f;                 // importance not clamped
        const float s2a = act ? s2 * sa : 0.f;
        g_pr += s1 * sa + s2a * dp;
        g_dp += s2a * pr;
        float* o = dlogits + (size_t)r * ld_g + (size_t)e * CH;
#pragma unroll
        for (int c = 0; c < C; ++c) { const float sg = tp[c]; o[c] = q[c] * sg * (1.f - sg) * obj_scale; }
        o[C] = (s1 * pr + (act ? s2 * pd : 0.f)) * sa * (1.f - sa) * alpha_scale;
    }
    const float cgx = -mult * ax, cgy = -mult * ay;          // d(source coord)/d(t) incl. the unnormalisation
    g_tx = wave_reduce_sum(g_tx) * cgx; g_ty = wave_reduce_sum(g_ty) * cgy;
    g_xs = wave_reduce_sum(g_xs) * cgx; g_ys = wave_reduce_sum(g_ys) * cgy;
    g_pr = wave_reduce_sum(g_pr); g_dp = wave_reduce_sum(g_dp);
    if (lane == 0) {
        *reinterpret_cast<float4*>(dnbox + (size_t)r * 4) = make_float4(2.f * g_tx, 2.f * g_ty, g_xs, g_ys);
        dpres[r] = g_pr;
        ddepth[r] = g_dp;
    }
}

}  // namespace

int render_num_blocks(int B, int I);

// sprites fp32 [N][ld_s] = [P*P][C+1]; x / recon [B][C][I][I]; aux: B*C*I*I float2 (dBCE/dpre / D, pre) or null
int render_fwd_c(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x, float* recon,
                 float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int ac, hipStream_t s) {
    if (C < 2 || C > RC_MAXC) return SPAIR_ERR_UNSUPPORTED;
    if (B <= 0 || HW <= 0 || I <= 0 || P <= 0 || ld_s < P * P * (C + 1)) return SPAIR_ERR_SHAPE;
    const dim3 grid(render_num_blocks(B, I));
    float2* a2 = reinterpret_cast<float2*>(aux);
    if (C == 2) hipLaunchKernelGGL(k_render_fwd_c<2>, grid, dim3(256), 0, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon, a2, bce_partial, B, HW, I, P, ac);
    else hipLaunchKernelGGL(k_render_fwd_c<3>, grid, dim3(256), 0, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon, a2, bce_partial, B, HW, I, P, ac);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// dlogits fp32 [N][ld_g] = [P*P][C+1]; dnbox [N][4]; dpres / ddepth [N]
int render_bwd_c(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
                 const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int C, int I, int P,
                 int ac, float obj_scale, float alpha_scale, hipStream_t s) {
    if (C < 2 || C > RC_MAXC) return SPAIR_ERR_UNSUPPORTED;
    if (B <= 0 || HW <= 0 || HW > 65535 || I <= 0 || P <= 0 || ld_s < P * P * (C + 1) || ld_g < P * P * (C + 1)) return SPAIR_ERR_SHAPE;
    const size_t lds = (size_t)P * P * (C + 2) * sizeof(float);
    if (lds > 64 * 1024) return SPAIR_ERR_UNSUPPORTED;
    const float2* a2 = reinterpret_cast<const float2*>(aux);
    if (C == 2) hipLaunchKernelGGL(k_render_bwd_c<2>, dim3(B, HW), dim3(64), lds, s, S, ld_s, nbox, pres, depth, ld_pd, a2, gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac, obj_scale, alpha_scale);
    else hipLaunchKernelGGL(k_render_bwd_c<3>, dim3(B, HW), dim3(64), lds, s, S, ld_s, nbox, pres, depth, ld_pd, a2, gloss, dlogits, dnbox, dpres, ddepth, ld_g, B, HW, I, P, ac, obj_scale, alpha_scale);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// unit-level C ABI (tests)
extern "C" int spair_render_fwd_rgb(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth, const float* x,
                                    float* recon, float* aux, float* bce_partial, int B, int HW, int C, int I, int P, int align_corners,
                                    void* stream) {
    return render_fwd_c(sprites, ld_s, nbox, pres, depth, 1, x, recon, aux, bce_partial, B, HW, C, I, P, align_corners, (hipStream_t)stream);
}
extern "C" int spair_render_bwd_rgb(const float* sprites, int ld_s, const float* nbox, const float* pres, const float* depth, const float* aux,
                                    const float* grad_loss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int B, int HW, int C,
                                    int I, int P, int align_corners, float obj_scale, float alpha_scale, void* stream) {
    return render_bwd_c(sprites, ld_s, nbox, pres, depth, 1, aux, grad_loss, dlogits, dnbox, dpres, ddepth, ld_s, B, HW, C, I, P, align_corners,
                        obj_scale, alpha_scale, (hipStream_t)stream);
}
