// K6 forward, third generation (reference: models.py:485-547, stn(inverse=True) modules.py:256-269): the inverse-STN sampling of the
// renderer on the MATRIX CORES.
//
// Bilinear sampling with zero padding is separable:  out_c[py][px] = sum_v sum_u Wy[py][v] . S_c[v][u] . Wx[px][u]  with the hat weights
// Wx[px][u] = max(0, 1 - |sx(px) - u|), Wy likewise (a tap on the padding simply has no column / row in S).  Per (object, 16 x 16 pixel
// tile) that is two small matrix products per channel (grey, alpha, importance):
//     T_c[v][px]    = sum_u S_c[v][u] . Wx[px][u]        2 x v_mfma_f32_16x16x32_f16   (32 sprite rows, K = the 32 texel columns)
//     out_c[py][px] = sum_v Wy[py][v] . T_c[v][px]       1 x v_mfma_f32_16x16x32_f16   (K = the 32 sprite rows)
// instead of k_render_fwd3's 4 LDS taps + ~41 VALU instructions per (pixel, object) pair.  What makes it cheap around the MFMAs:
//   * the sprite is fp16 (grey, alpha) pairs already (what the decoder writes), and the A operand of the first product is a plain row
//     piece of it: lane (row v, k-group q) needs texels 8q .. 8q+7 of row v = 32 contiguous bytes -- two 16-byte loads per lane straight
//     from L2 into registers, de-interleaved by v_perm_b32.  No LDS staging, no direct-to-LDS DMA, no tap tables;
//   * the rows of a sprite a tile can touch form a window [v0, v1]: the 32 rows are v0 .. v0+31 (rows >= P read zeros through the
//     buffer descriptor's range check), and when v1 - v0 < 16 -- about half of the (object, tile) pairs -- the second 16-row tile
//     (its loads, its three MFMAs) is skipped; v0 and that flag are found once by the thread that culls the object;
//   * the accumulator of the first product IS the B operand of the second up to the fp16 conversion: lane (px, q) holds rows 4q..4q+3
//     of both 16-row tiles, and K is only a summation index, so the second product simply enumerates the sprite rows in that order
//     (the hat weights Wy are built in the same order);
//   * the hat weights are formed in fp32 and rounded to fp16 once (render3.h): the small weight of a tap pair keeps 11 significant
//     bits, the pair sums to 1 +- 2^-12;
//   * per-object parameters come from two per-object records written once per step by k_render_prep (48 bytes together): the cull
//     record (pixel footprint + the row coefficients: the tile cull is four integer compares) and the object record (source-coordinate
//     coefficients, presence, importance scale and floor); a wave keeps the records of its
//     objects one per lane and broadcasts a field with v_readlane;
//   * importance max(alpha * pd, 0.01) = pd * max(alpha, 0.01 / pd): one v_pk_max_f16 per two texels, pd applied to the sampled tile
//     in fp32 (pd <= 0.01: the importance is the constant 0.01 = 0.01 * max(alpha, 1)).
// Work split: one workgroup (4 waves) per (sample, 16 x 16 tile); the tile's surviving objects are dealt round-robin to the waves, each
// wave composites whole objects into its own (num, den) tile in registers, and the four partial tiles meet in LDS at the end.
//
// Numerics (fp16 sprites in both kernels, so this is about the sampling only): hat weights and the x-interpolated rows T are rounded to
// fp16 once each (nearest even, 2^-12 relative), the importance floor 0.01 / pd is an fp16 number; the products are exact and accumulate
// in fp32.  Measured against the oracle on the same fp16 sprites: tests/test_kernels_gpu.py::test_render16m_fwd_vs_oracle.
#include <stdlib.h>
#include "render3.h"

template <int AC, int IP2>
__global__ __launch_bounds__(256) void k_render_prep(const float* __restrict__ nbox, const float* __restrict__ pres,
                                                     const float* __restrict__ depth, int ld_pd, RenderObjRec* __restrict__ orec,
                                                     RenderCullRec* __restrict__ crec, RenderBwdRec* __restrict__ brec, int B, int HW, int I,
                                                     int P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // = b * HW + k
    if (idx >= B * HW) return;
    const int b = idx / HW, k = idx - b * HW;
    const int r = k * B + b;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    const float pr = pres[(size_t)r * ld_pd], pd = pr * depth[(size_t)r * ld_pd];
    // the same expressions as the backward kernels for the footprint: exact w.r.t. the coordinates the gradients are taken at
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    float ax = 1.f / nb.z, bx = -tx / nb.z, ay = 1.f / nb.w, by = -ty / nb.w;
    const float inv_I = 1.f / (float)I;
    int x0 = R3_EMPTY, x1 = 0, y0 = R3_EMPTY, y1 = 0;
    const float big = 1e30f;
    if (fabsf(ax) < big && fabsf(bx) < big && fabsf(ay) < big && fabsf(by) < big && ax > 0.f && ay > 0.f) {
        float sx0, sxa, sy0, sya;
        src_affine(ax, bx, I, P, AC, sx0, sxa);
        src_affine(ay, by, I, P, AC, sy0, sya);
        int lo, hi;
        rb2_range<AC, IP2>(ax, bx, sx0, __builtin_amdgcn_rcpf(sxa), I, inv_I, P, lo, hi);
        if (lo <= hi) { x0 = lo; x1 = hi; }
        rb2_range<AC, IP2>(ay, by, sy0, __builtin_amdgcn_rcpf(sya), I, inv_I, P, lo, hi);
        if (lo <= hi) { y0 = lo; y1 = hi; }
    } else {
        ax = bx = ay = by = 0.f;
    }
    const float hp = 0.5f * (float)P;
    RenderObjRec o;
    o.Ax = ax * hp; o.Ay = ay * hp;
    const float bcx = fmaf(bx + 1.f, hp, -0.5f), bcy = fmaf(by + 1.f, hp, -0.5f);
    o.Bx = bcx; o.By = bcy;
    o.pres = pr;
    // max(alpha * pd, 0.01) = mscale * max(alpha, mfloor)
    const bool flat = !(pd > 0.01f);                          // alpha <= 1: the importance is 0.01 everywhere on the sprite
    o.mscale = flat ? 0.01f : pd;
    const _Float16 fl = flat ? (_Float16)1.f : (_Float16)fmaxf(0.01f / pd, 6.2e-5f);
    const r3_h2 fl2 = {fl, fl};
    o.mfloor = __builtin_bit_cast(unsigned, fl2);
    o.depth = depth[(size_t)r * ld_pd];
    RenderCullRec c;
    c.Ay = o.Ay; c.By = o.By;
    c.xr = (unsigned)x0 | ((unsigned)x1 << 16);
    c.yr = (unsigned)y0 | ((unsigned)y1 << 16);
    orec[idx] = o;
    crec[idx] = c;
    brec[idx] = RenderBwdRec{ax, bx, ay, by};
}

struct R3Obj {                      // everything of one (object, tile) pair that is in flight before its arithmetic
    R3Frag t0, t1;
    float Ax, Bx, Ay, By, pres, mscale;         // the object record (wave-uniform)
    unsigned mfloor;
    unsigned e;                     // list entry: k | v0 << 16 | two << 24
};

template <int IP2, int NT>
__global__ __launch_bounds__(256) void k_render_fwd_mma(const void* __restrict__ S, unsigned s_bytes, const RenderObjRec* __restrict__ orec,
                                                        const RenderCullRec* __restrict__ crec, const float* __restrict__ x,
                                                        float* __restrict__ recon, float2* __restrict__ aux,
                                                        float* __restrict__ bce_partial, int B, int HW, int I) {
    __shared__ unsigned list[4][R3_MAXHW / 4];
    __shared__ int cnt[4];
    __shared__ float red[4][NT][8][64];
    __shared__ float red4[4];
    constexpr int P = R3_P;
    constexpr int RH = RT * NT;                 // pixel rows of the region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv_I = 1.f / (float)I;
    // grid (nx, columns * regions, B / nx), nx = 8 when B % 8 == 0: workgroups are dealt round-robin over the 8 XCDs in linear order (x
    // fastest), so every region of sample b = z * nx + x runs on XCD x -- its sprites are fetched from HBM once and then hit that XCD's L2
    const int tiles_x = (I + RT - 1) / RT;
    const int b = blockIdx.z * gridDim.x + blockIdx.x, reg = blockIdx.y;
    const int ryi = IP2 ? reg >> (31 - __builtin_clz(tiles_x)) : reg / tiles_x;
    const int tx0 = (reg - ryi * tiles_x) * RT, ty0 = ryi * RH;
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RH, I) - 1;
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int l15 = lane & 15, q = lane >> 4;
    // this thread's pixels in the epilogue (accumulator layout of the second product: column px, rows 4q .. 4q+3; wave w finishes row 4q+w
    // of each of the region's NT tiles)
    const int px = tx0 + l15, py = ty0 + 4 * q + wave;
    size_t pi[NT];
    float xv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        pi[t] = ((size_t)b * I + min(py + RT * t, I - 1)) * I + min(px, I - 1);
        xv[t] = x[pi[t]];
    }
    const RenderObjRec* orecb = orec + (size_t)b * HW;
    const RenderCullRec* crecb = crec + (size_t)b * HW;

    // ---- 1. cull: footprint against the region; which of its NT tiles a hit reaches; the sprite-row window [v0, v1] of the hit (its
    // first 16-row tile starts at min(v0, P - 16), so that tile never runs past the sprite; the second one is needed when the window is
    // longer than 16 rows)
    int nw = 0;
    for (int k0 = 0; k0 < HW; k0 += 256) {
        const int k = k0 + tid;
        bool hit = false;
        unsigned e = 0;
        if (k < HW) {
            const uint4 rc = *reinterpret_cast<const uint4*>(crecb + k);
            const int x0 = rc.z & 0xffff, x1 = rc.z >> 16, y0 = rc.w & 0xffff, y1 = rc.w >> 16;
            hit = x0 <= tx1 && x1 >= tx0 && y0 <= ty1 && y1 >= ty0;
            const float ay = __uint_as_float(rc.x), by = __uint_as_float(rc.y);
            // the coordinates of the first / last pixel row that can draw: their taps are floor and floor + 1
            const float s0 = fmaf(ay, rf_base<0, IP2>(max(ty0, y0), I, inv_I), by);
            const float s1 = fmaf(ay, rf_base<0, IP2>(min(ty1, y1), I, inv_I), by);
            const int v0 = min(max((int)floorf(s0), 0), P - 16), v1 = max(min((int)floorf(s1) + 1, P - 1), v0);
            unsigned ym = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) ym |= (y0 <= ty0 + RT * t + RT - 1 && y1 >= ty0 + RT * t) ? (1u << t) : 0u;
            e = (unsigned)k | ((unsigned)v0 << 16) | ((v1 - v0) >= 16 ? (1u << 24) : 0u) | (ym << 25);
        }
        const unsigned long long bal = __ballot(hit);
        if (hit) list[wave][nw + __popcll(bal & ((1ull << lane) - 1ull))] = e;
        nw += __popcll(bal);
    }
    if (lane == 0) cnt[wave] = nw;
    __syncthreads();
    const int c0 = cnt[0], c1 = c0 + cnt[1], c2 = c1 + cnt[2], nall = c2 + cnt[3];

    // ---- per-lane constants of the operand fragments
    // first product, B operand Wx: lane (column px = l15, k-group q) holds texel columns u = 8q + j; u >= P stays empty
    // second product, A operand Wy: lane (row py = l15, k-group q) holds sprite rows v0 + {4q .. 4q+3, 16+4q .. 16+4q+3}
    r3_f2 cx[4], cy[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const int u = 8 * q + 2 * jp;
        cx[jp] = r3_f2{u < P ? -(float)u : -1.0e4f, u + 1 < P ? -(float)(u + 1) : -1.0e4f};
        const int v = (jp < 2 ? 4 * q + 2 * jp : 16 + 4 * q + 2 * (jp - 2));
        cy[jp] = r3_f2{-(float)v, -(float)(v + 1)};
    }
    const float basex = rf_base<0, IP2>(min(tx0 + l15, I - 1), I, inv_I);
    float basey[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) basey[t] = rf_base<0, IP2>(min(ty0 + RT * t + l15, I - 1), I, inv_I);
    // first product, A operand: lane (sprite row l15 of the 16-row tile, k-group q): bytes 32q .. 32q+31 of the row
    // (the last k-group's second half would be texels 28 .. 31: it re-reads the first half, its weights are empty)
    const unsigned voff = (unsigned)(l15 * R3_ROWB + 32 * q), voffh = voff + (q == 3 ? 0u : 16u);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(S), (short)0, (int)s_bytes, 0x00020000);

    f32x4 num[NT], den[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { num[t] = f32x4{0.f, 0.f, 0.f, 0.f}; den[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // this wave's objects: entries wave, wave + 4, ... of the concatenated lists; 64 of them at a time, one per lane
    for (int i0 = wave; i0 < nall; i0 += 256) {
        const int i = i0 + 4 * lane;
        unsigned mine = 0;
        if (i < nall) {
            const int seg = (i >= c0) + (i >= c1) + (i >= c2);
            const int base = seg == 0 ? 0 : seg == 1 ? c0 : seg == 2 ? c1 : c2;
            mine = list[seg][i - base];
        }
        const int n = min(64, (nall - i0 + 3) >> 2);
        // lane j holds the record of this wave's j-th object (one vector load per 64 objects; a record fetched by scalar loads inside the
        // object loop put their latency on every object)
        const uint4 myra = reinterpret_cast<const uint4*>(orecb + (mine & 0xffff))[0];
        const uint4 myrb = reinterpret_cast<const uint4*>(orecb + (mine & 0xffff))[1];
        auto rl = [&](unsigned v, int j) { return (unsigned)__builtin_amdgcn_readlane((int)v, j); };
        auto fetch = [&](int j) {
            R3Obj o;
            const int jj = min(j, n - 1);
            o.e = rl(mine, jj);
            o.Ax = __uint_as_float(rl(myra.x, jj)); o.Bx = __uint_as_float(rl(myra.y, jj));
            o.Ay = __uint_as_float(rl(myra.z, jj)); o.By = __uint_as_float(rl(myra.w, jj));
            o.pres = __uint_as_float(rl(myrb.x, jj)); o.mscale = __uint_as_float(rl(myrb.y, jj));
            o.mfloor = rl(myrb.z, jj);
            const int k = o.e & 0xffff;
            const unsigned v0 = (o.e >> 16) & 0xff;
            const unsigned ob = (unsigned)(k * B + b) * (unsigned)R3_SPRB + v0 * (unsigned)R3_ROWB;
            // rows of the second tile beyond the sprite read zeros through the descriptor's range check (they carry weight: they must BE zero)
            const bool ok1 = (v0 + (unsigned)l15 + 16u < (unsigned)P) & (((o.e >> 24) & 1u) != 0);
            const unsigned o1l = ok1 ? ob + voff + 16u * R3_ROWB : BUF_OOB, o1h = ok1 ? ob + voffh + 16u * R3_ROWB : BUF_OOB;
            o.t0.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(ob + voff), 0, 0);
            o.t0.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)(ob + voffh), 0, 0);
            o.t1.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o1l, 0, 0);
            o.t1.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o1h, 0, 0);
            return o;
        };
        auto composite = [&](const R3Obj& o) {
            const unsigned v0 = (o.e >> 16) & 0xff;
            const float sx = fmaf(o.Ax, basex, o.Bx);
            const r3_h8 wx = r3_hat8(sx, cx);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            r3_h8 sg, sa, sm;
            r3_split(o.t0, o.mfloor, sg, sa, sm);
            const f32x4 tg0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sg, wx, z, 0, 0, 0);
            const f32x4 ta0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa, wx, z, 0, 0, 0);
            const f32x4 tm0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sm, wx, z, 0, 0, 0);
            u32x4_t hg = {r3_pk(tg0[0], tg0[1]), r3_pk(tg0[2], tg0[3]), 0u, 0u};
            u32x4_t ha = {r3_pk(ta0[0], ta0[1]), r3_pk(ta0[2], ta0[3]), 0u, 0u};
            u32x4_t hm = {r3_pk(tm0[0], tm0[1]), r3_pk(tm0[2], tm0[3]), 0u, 0u};
            if ((o.e >> 24) & 1u) {
                r3_split(o.t1, v0 + (unsigned)l15 + 16u < (unsigned)P ? o.mfloor : 0u, sg, sa, sm);
                const f32x4 tg1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sg, wx, z, 0, 0, 0);
                const f32x4 ta1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa, wx, z, 0, 0, 0);
                const f32x4 tm1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sm, wx, z, 0, 0, 0);
                hg[2] = r3_pk(tg1[0], tg1[1]); hg[3] = r3_pk(tg1[2], tg1[3]);
                ha[2] = r3_pk(ta1[0], ta1[1]); ha[3] = r3_pk(ta1[2], ta1[3]);
                hm[2] = r3_pk(tm1[0], tm1[1]); hm[3] = r3_pk(tm1[2], tm1[3]);
            }
            const float yoff = (float)v0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (NT == 1 || ((o.e >> (25 + t)) & 1u)) {
                    const float sy = fmaf(o.Ay, basey[t], o.By) - yoff;                 // ... relative to the window's first row
                    const r3_h8 wy = r3_hat8(sy, cy);
                    const f32x4 og = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, hg), z, 0, 0, 0);
                    const f32x4 oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, ha), z, 0, 0, 0);
                    const f32x4 om = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, hm), z, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = oa[r] * o.pres, m = om[r] * o.mscale;
                        num[t][r] = fmaf(og[r] * a, m + 1e-9f, num[t][r]);
                        den[t][r] += m;
                    }
                }
            }
        };
        R3Obj oa_ = fetch(0);
        for (int j = 0; j < n; j += 2) {
            const R3Obj ob_ = fetch(j + 1);
            composite(oa_);
            oa_ = fetch(j + 2);
            if (j + 1 < n) composite(ob_);
        }
    }
    // ---- the four waves' partial tiles
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[wave][t][r][lane] = num[t][r]; red[wave][t][4 + r][lane] = den[t][r]; }
    __syncthreads();
    float bce = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float nsum = (red[0][t][wave][lane] + red[1][t][wave][lane]) + (red[2][t][wave][lane] + red[3][t][wave][lane]);
        const float dsum = (red[0][t][4 + wave][lane] + red[1][t][4 + wave][lane]) + (red[2][t][4 + wave][lane] + red[3][t][4 + wave][lane]);
        if (px < I && py + RT * t < I) {
            const float D = dsum + (float)HW * 1e-9f;   // every object adds 1e-9 (models.py:527)
            const float invD = r3_rcp(D);
            const float pre = nsum * invD;
            const float r = fminf(fmaxf(pre, 0.f), 1.f);
            recon[pi[t]] = r;
            // torch BCE: log clamped at -100; backward denominator max(r(1-r), 1e-12)
            bce -= xv[t] * fmaxf(logf(r), -100.f) + (1.f - xv[t]) * fmaxf(logf(1.f - r), -100.f);
            if (aux) {
                const float gr = (pre >= 0.f && pre <= 1.f) ? (r - xv[t]) * r3_rcp(fmaxf(r * (1.f - r), 1e-12f)) : 0.f;
                aux[pi[t]] = make_float2(gr * invD, pre);
            }
        }
    }
    bce = block_reduce_sum_256(bce, red4);
    if (tid < NT) bce_partial[wg * NT + tid] = tid == 0 ? bce : 0.f;
}

int render_prep_bytes(int B, int HW) { return B * HW * R3_REC_BYTES; }
// what the backward kernel reads of the records (render2.hip): footprints and raw parameters
const void* render_rec_cull(const void* rec, int B, int HW) { return reinterpret_cast<const RenderObjRec*>(rec) + (size_t)B * HW; }
const void* render_rec_bwd(const void* rec, int B, int HW) {
    return reinterpret_cast<const RenderCullRec*>(render_rec_cull(rec, B, HW)) + (size_t)B * HW;
}
int render_prep_supported(int HW, int I, int P, int ac) { return !ac && P == R3_P && HW <= R3_MAXHW && I < (int)R3_EMPTY; }

// SPAIR_ERR_UNSUPPORTED: the caller keeps k_render_fwd3 (which needs no records)
int render_prep(const float* nbox, const float* pres, const float* depth, int ld_pd, void* rec, int B, int HW, int I, int P, int ac,
                hipStream_t s) {
    if (!render_prep_supported(HW, I, P, ac) || (reinterpret_cast<uintptr_t>(rec) & 15)) return SPAIR_ERR_UNSUPPORTED;
    const dim3 grid((B * HW + 255) / 256), block(256);
    RenderObjRec* orec = reinterpret_cast<RenderObjRec*>(rec);
    RenderCullRec* crec = reinterpret_cast<RenderCullRec*>(orec + (size_t)B * HW);
    RenderBwdRec* brec = reinterpret_cast<RenderBwdRec*>(crec + (size_t)B * HW);
    if ((I & (I - 1)) == 0)
        hipLaunchKernelGGL((k_render_prep<0, 1>), grid, block, 0, s, nbox, pres, depth, ld_pd, orec, crec, brec, B, HW, I, P);
    else
        hipLaunchKernelGGL((k_render_prep<0, 0>), grid, block, 0, s, nbox, pres, depth, ld_pd, orec, crec, brec, B, HW, I, P);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

#ifndef R3_NT
#define R3_NT 4                     // 16-row tiles per workgroup region (one tile column, R3_NT tiles tall)
#endif
int render_fwd_mma(const void* S16, int ld_s, const void* rec, const float* x, float* recon, float* aux, float* bce_partial, int B, int HW,
                   int I, int P, int ac, hipStream_t s) {
    if (ac || P != R3_P || ld_s != R3_P * R3_P * 2 || HW > R3_MAXHW || I >= (int)R3_EMPTY) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * R3_SPRB >= 0xfffffff0ull - 64) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S16) & 15) || (reinterpret_cast<uintptr_t>(rec) & 15)) return SPAIR_ERR_UNSUPPORTED;
    const int t = (I + RT - 1) / RT, nx = (B & 7) == 0 ? 8 : 1;
    // regions of R3_NT tiles when they tile the image's tile rows exactly (the bce_partial slots are the tiles'), single tiles otherwise
    const int nt = (t % R3_NT) == 0 ? R3_NT : 1;
    if (t * (t / nt) > 65535 || B / nx > 65535) return SPAIR_ERR_UNSUPPORTED;
    const dim3 grid(nx, t * (t / nt), B / nx), block(256);
    const unsigned s_bytes = (unsigned)((size_t)B * HW * R3_SPRB);
    const RenderObjRec* orec = reinterpret_cast<const RenderObjRec*>(rec);
    const RenderCullRec* crec = reinterpret_cast<const RenderCullRec*>(orec + (size_t)B * HW);
    const bool ip2 = (I & (I - 1)) == 0;
#define R3_LAUNCH(IP2_, NT_)                                                                                                           \
    hipLaunchKernelGGL((k_render_fwd_mma<IP2_, NT_>), grid, block, 0, s, S16, s_bytes, orec, crec, x, recon, reinterpret_cast<float2*>(aux), \
                       bce_partial, B, HW, I)
    if (nt == 1) { if (ip2) R3_LAUNCH(1, 1); else R3_LAUNCH(0, 1); }
    else { if (ip2) R3_LAUNCH(1, R3_NT); else R3_LAUNCH(0, R3_NT); }
#undef R3_LAUNCH
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
