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
//   * the hat weights are exact in fp16: the source coordinate is rounded to a multiple of 2^-11 texel, so 1 - f and f have 11
//     significant bits and the pair still sums to exactly 1;
//   * per-object parameters (exact inverse affine, presence, presence * depth, pixel footprint) come from a 32-byte per-object record
//     written once per step by k_render_prep: the tile cull is four integer compares, and a wave fetches the record of the object it
//     works on with scalar loads.
// Work split: one workgroup (4 waves) per (sample, 16 x 16 tile); the tile's surviving objects are dealt round-robin to the waves, each
// wave composites whole objects into its own (num, den) tile in registers, and the four partial tiles meet in LDS at the end.
//
// Numerics (fp16 sprites in both kernels, so this is about the sampling only): the coordinate rounding moves a sample by <= 2^-12 texel,
// T is rounded to fp16 once (RNE, 2^-12 relative), the importance max(alpha * pres * depth, 0.01) is formed in fp16; the products are exact
// and accumulate in fp32.  Measured against the oracle on the same fp16 sprites: tests/test_kernels_gpu.py::test_render16m_fwd_vs_oracle.
#include <stdlib.h>
#include "render_common.h"

typedef _Float16 r3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 r3_h2 __attribute__((ext_vector_type(2)));
typedef float r3_f2 __attribute__((ext_vector_type(2)));

#define R3_P 28                     // sprite side this kernel is built for (row = 112 B = 7 x 16 B)
#define R3_ROWB (R3_P * 4)
#define R3_SPRB (R3_P * R3_P * 4)
#define R3_MAXHW 1024               // objects per sample (list capacity: HW / 4 entries per culling wave)
#define R3_EMPTY 0x7fffu            // first index of an empty footprint
#define R3_QMAGIC 6144.0f           // 1.5 * 2^12: (s + M) - M rounds s to a multiple of 2^-11 for |s| < 2^11

// 32 bytes per object, sample-major [b][k]
struct __attribute__((aligned(16))) RenderRec {
    float ax, bx, pres;
    unsigned pdh;                   // presence * depth as an fp16 pair (both halves)
    float ay, by;
    unsigned xr, yr;                // pixel footprint: first | last << 16 (first = R3_EMPTY: nothing to draw)
};

template <int AC, int IP2>
__global__ __launch_bounds__(256) void k_render_prep(const float* __restrict__ nbox, const float* __restrict__ pres,
                                                     const float* __restrict__ depth, int ld_pd, RenderRec* __restrict__ rec, int B, int HW,
                                                     int I, int P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // = b * HW + k
    if (idx >= B * HW) return;
    const int b = idx / HW, k = idx - b * HW;
    const int r = k * B + b;
    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
    const float pr = pres[(size_t)r * ld_pd], pd = pr * depth[(size_t)r * ld_pd];
    // the same expressions as the backward kernels: forward and backward must agree on the coordinates to the bit
    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
    RenderRec o;
    o.ax = 1.f / nb.z; o.bx = -tx / nb.z; o.ay = 1.f / nb.w; o.by = -ty / nb.w;
    o.pres = pr;
    const r3_h2 ph = {(_Float16)pd, (_Float16)pd};
    o.pdh = __builtin_bit_cast(unsigned, ph);
    const float inv_I = 1.f / (float)I;
    int x0 = R3_EMPTY, x1 = 0, y0 = R3_EMPTY, y1 = 0;
    const float big = 1e30f;
    if (fabsf(o.ax) < big && fabsf(o.bx) < big && fabsf(o.ay) < big && fabsf(o.by) < big && o.ax > 0.f && o.ay > 0.f) {
        float sx0, sxa, sy0, sya;
        src_affine(o.ax, o.bx, I, P, AC, sx0, sxa);
        src_affine(o.ay, o.by, I, P, AC, sy0, sya);
        int lo, hi;
        rb2_range<AC, IP2>(o.ax, o.bx, sx0, __builtin_amdgcn_rcpf(sxa), I, inv_I, P, lo, hi);
        if (lo <= hi) { x0 = lo; x1 = hi; }
        rb2_range<AC, IP2>(o.ay, o.by, sy0, __builtin_amdgcn_rcpf(sya), I, inv_I, P, lo, hi);
        if (lo <= hi) { y0 = lo; y1 = hi; }
    } else {
        o.ax = o.bx = o.ay = o.by = 0.f;
    }
    o.xr = (unsigned)x0 | ((unsigned)x1 << 16);
    o.yr = (unsigned)y0 | ((unsigned)y1 << 16);
    rec[idx] = o;
}

// 8 hat weights max(0, 1 - |s - u_j|) of a coordinate that is a multiple of 2^-11, as an fp16 MFMA fragment.  c[jp] = -(u_2jp, u_2jp+1)
// (a slot that must stay empty carries a large value).  d = s - u is exact in fp32 and, where |d| < 1, in fp16; 1 - |d| likewise.
__device__ __forceinline__ r3_h8 r3_hat8(float s, const r3_f2 (&c)[4]) {
    const r3_h2 one = {(_Float16)1.f, (_Float16)1.f}, zero = {(_Float16)0.f, (_Float16)0.f};
    const r3_f2 s2 = {s, s};
    r3_h2 w[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        const r3_f2 d = s2 + c[jp];
        const r3_h2 dh = __builtin_bit_cast(r3_h2, __builtin_amdgcn_cvt_pkrtz(d.x, d.y));
        const r3_h2 ad = __builtin_elementwise_max(dh, -dh);
        w[jp] = __builtin_elementwise_max(one - ad, zero);
    }
    return r3_h8{w[0].x, w[0].y, w[1].x, w[1].y, w[2].x, w[2].y, w[3].x, w[3].y};
}

struct R3Frag { u32x4_t lo, hi; };      // 8 texels (grey, alpha) of one sprite row

// fl: the importance floor 0.01 as an fp16 pair -- 0 for a sprite row beyond the sprite, whose texels read as zeros: the PADDING's importance
// is 0, not the floor (the reference clamps the importance sprite, then grid_sample pads it with zeros)
__device__ __forceinline__ void r3_split(const R3Frag& f, unsigned pdh, unsigned fl, r3_h8& g, r3_h8& a, r3_h8& m) {
    const unsigned d[8] = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
    const r3_h2 pd2 = __builtin_bit_cast(r3_h2, pdh);
    const r3_h2 floor2 = __builtin_bit_cast(r3_h2, fl);
    unsigned gg[4], aa[4], mm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gg[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
        aa[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
        const r3_h2 im = __builtin_elementwise_max(__builtin_bit_cast(r3_h2, aa[i]) * pd2, floor2);     // importance, models.py:497-499
        mm[i] = __builtin_bit_cast(unsigned, im);
    }
    g = __builtin_bit_cast(r3_h8, u32x4_t{gg[0], gg[1], gg[2], gg[3]});
    a = __builtin_bit_cast(r3_h8, u32x4_t{aa[0], aa[1], aa[2], aa[3]});
    m = __builtin_bit_cast(r3_h8, u32x4_t{mm[0], mm[1], mm[2], mm[3]});
}

__device__ __forceinline__ unsigned r3_pk(float a, float b) {
    const r3_h2 h = {(_Float16)a, (_Float16)b};            // v_cvt_pk_f16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, h);
}

struct R3Obj {                      // everything of one (object, tile) pair that is in flight before its arithmetic
    R3Frag t0, t1;
    uint4 ra, rb;                   // the record (wave-uniform)
    unsigned e;                     // list entry: k | v0 << 16 | two << 24
};

template <int IP2>
__global__ __launch_bounds__(256) void k_render_fwd_mma(const void* __restrict__ S, unsigned s_bytes, const RenderRec* __restrict__ rec,
                                                        const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                        float* __restrict__ bce_partial, int B, int HW, int I) {
    __shared__ unsigned list[4][R3_MAXHW / 4];
    __shared__ int cnt[4];
    __shared__ float red[4][8][64];
    __shared__ float red4[4];
    constexpr int P = R3_P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv_I = 1.f / (float)I;
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
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
    const int l15 = lane & 15, q = lane >> 4;
    // this thread's pixel in the epilogue (accumulator layout of the second product: column px, rows 4q .. 4q+3; wave w finishes row 4q+w)
    const int px = tx0 + l15, py = ty0 + 4 * q + wave;
    const bool inside = px < I && py < I;
    const size_t pi = ((size_t)b * I + min(py, I - 1)) * I + min(px, I - 1);
    const float xv = x[pi];
    const RenderRec* recb = rec + (size_t)b * HW;

    // ---- 1. cull: footprint against the tile, the sprite-row window of the hits
    int nw = 0;
    for (int k0 = 0; k0 < HW; k0 += 256) {
        const int k = k0 + tid;
        bool hit = false;
        unsigned e = 0;
        if (k < HW) {
            const uint4 rb = reinterpret_cast<const uint4*>(recb + k)[1];
            const int x0 = rb.z & 0xffff, x1 = rb.z >> 16, y0 = rb.w & 0xffff, y1 = rb.w >> 16;
            hit = x0 <= tx1 && x1 >= tx0 && y0 <= ty1 && y1 >= ty0;
            float g;
            const float ay = __uint_as_float(rb.x), by = __uint_as_float(rb.y);
            const float s0 = src_from_base(ay, by, rf_base<0, IP2>(max(ty0, y0), I, inv_I), P, 0, g);
            const float s1 = src_from_base(ay, by, rf_base<0, IP2>(min(ty1, y1), I, inv_I), P, 0, g);
            const int v0 = min(max((int)floorf(s0), 0), P - 1), v1 = max(min((int)floorf(s1) + 1, P - 1), v0);
            e = (unsigned)k | ((unsigned)v0 << 16) | ((v1 - v0) >= 16 ? (1u << 24) : 0u);
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
    const float basey = rf_base<0, IP2>(min(ty0 + l15, I - 1), I, inv_I);
    // first product, A operand: lane (sprite row l15 of the 16-row tile, k-group q): bytes 32q .. 32q+31 of the row
    // (the last k-group's second half would be texels 28 .. 31: it re-reads the first half, its weights are empty)
    const unsigned voff = (unsigned)(l15 * R3_ROWB + 32 * q), voffh = voff + (q == 3 ? 0u : 16u);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(S), (short)0, (int)s_bytes, 0x00020000);

    const r3_h2 floor_v = {(_Float16)0.01f, (_Float16)0.01f};
    const unsigned floor_h2 = __builtin_bit_cast(unsigned, floor_v);
    f32x4 num = {0.f, 0.f, 0.f, 0.f}, den = {0.f, 0.f, 0.f, 0.f};
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
        auto fetch = [&](int j) {
            R3Obj o;
            o.e = (unsigned)__builtin_amdgcn_readlane((int)mine, min(j, n - 1));
            const int k = o.e & 0xffff;
            const unsigned v0 = (o.e >> 16) & 0xff;
            const uint4* rp = reinterpret_cast<const uint4*>(recb + k);
            o.ra = rp[0]; o.rb = rp[1];
            const unsigned ob = (unsigned)(k * B + b) * (unsigned)R3_SPRB + v0 * (unsigned)R3_ROWB;
            const unsigned row0 = v0 + (unsigned)l15;
            const bool ok0 = row0 < (unsigned)P, ok1 = (row0 + 16u < (unsigned)P) & ((o.e >> 24) != 0);
            // masked lanes read zeros through the descriptor's range check (rows >= P carry weight, so they must BE zero)
            const unsigned o0l = ok0 ? ob + voff : BUF_OOB, o0h = ok0 ? ob + voffh : BUF_OOB;
            const unsigned o1l = ok1 ? ob + voff + 16u * R3_ROWB : BUF_OOB, o1h = ok1 ? ob + voffh + 16u * R3_ROWB : BUF_OOB;
            o.t0.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o0l, 0, 0);
            o.t0.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o0h, 0, 0);
            o.t1.lo = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o1l, 0, 0);
            o.t1.hi = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)o1h, 0, 0);
            return o;
        };
        auto composite = [&](const R3Obj& o) {
            const float ax = __uint_as_float(o.ra.x), bx = __uint_as_float(o.ra.y), pr = __uint_as_float(o.ra.z);
            const float ay = __uint_as_float(o.rb.x), by = __uint_as_float(o.rb.y);
            const unsigned pdh = o.ra.w;
            const float v0f = (float)((o.e >> 16) & 0xff);
            float g;
            const float sx = (src_from_base(ax, bx, basex, P, 0, g) + R3_QMAGIC) - R3_QMAGIC;
            const float sy = ((src_from_base(ay, by, basey, P, 0, g) + R3_QMAGIC) - R3_QMAGIC) - v0f;
            const r3_h8 wx = r3_hat8(sx, cx), wy = r3_hat8(sy, cy);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            r3_h8 sg, sa, sm;
            const unsigned row0 = ((o.e >> 16) & 0xff) + (unsigned)l15;
            r3_split(o.t0, pdh, row0 < (unsigned)P ? floor_h2 : 0u, sg, sa, sm);
            const f32x4 tg0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sg, wx, z, 0, 0, 0);
            const f32x4 ta0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa, wx, z, 0, 0, 0);
            const f32x4 tm0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sm, wx, z, 0, 0, 0);
            u32x4_t hg = {r3_pk(tg0[0], tg0[1]), r3_pk(tg0[2], tg0[3]), 0u, 0u};
            u32x4_t ha = {r3_pk(ta0[0], ta0[1]), r3_pk(ta0[2], ta0[3]), 0u, 0u};
            u32x4_t hm = {r3_pk(tm0[0], tm0[1]), r3_pk(tm0[2], tm0[3]), 0u, 0u};
            if (o.e >> 24) {
                r3_split(o.t1, pdh, row0 + 16u < (unsigned)P ? floor_h2 : 0u, sg, sa, sm);
                const f32x4 tg1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sg, wx, z, 0, 0, 0);
                const f32x4 ta1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa, wx, z, 0, 0, 0);
                const f32x4 tm1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(sm, wx, z, 0, 0, 0);
                hg[2] = r3_pk(tg1[0], tg1[1]); hg[3] = r3_pk(tg1[2], tg1[3]);
                ha[2] = r3_pk(ta1[0], ta1[1]); ha[3] = r3_pk(ta1[2], ta1[3]);
                hm[2] = r3_pk(tm1[0], tm1[1]); hm[3] = r3_pk(tm1[2], tm1[3]);
            }
            const f32x4 og = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, hg), z, 0, 0, 0);
            const f32x4 oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, ha), z, 0, 0, 0);
            const f32x4 om = __builtin_amdgcn_mfma_f32_16x16x32_f16(wy, __builtin_bit_cast(r3_h8, hm), z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = oa[r] * pr;
                num[r] = fmaf(og[r] * a, om[r] + 1e-9f, num[r]);
                den[r] += om[r];
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
    for (int r = 0; r < 4; ++r) { red[wave][r][lane] = num[r]; red[wave][4 + r][lane] = den[r]; }
    __syncthreads();
    const float nsum = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
    const float dsum = (red[0][4 + wave][lane] + red[1][4 + wave][lane]) + (red[2][4 + wave][lane] + red[3][4 + wave][lane]);
    float bce = 0.f;
    if (inside) {
        const float D = dsum + (float)HW * 1e-9f;   // every object adds 1e-9 (models.py:527)
        const float invD = 1.f / D;
        const float pre = nsum * invD;
        const float r = fminf(fmaxf(pre, 0.f), 1.f);
        recon[pi] = r;
        // torch BCE: log clamped at -100; backward denominator max(r(1-r), 1e-12)
        bce = -(xv * fmaxf(logf(r), -100.f) + (1.f - xv) * fmaxf(logf(1.f - r), -100.f));
        if (aux) {
            const float gr = (pre >= 0.f && pre <= 1.f) ? (r - xv) / fmaxf(r * (1.f - r), 1e-12f) : 0.f;
            aux[pi] = make_float2(gr * invD, pre);
        }
    }
    bce = block_reduce_sum_256(bce, red4);
    if (tid == 0) bce_partial[blockIdx.x] = bce;
}

int render_prep_bytes(int B, int HW) { return B * HW * (int)sizeof(RenderRec); }

// SPAIR_ERR_UNSUPPORTED: the caller keeps k_render_fwd3 (which needs no records)
int render_prep(const float* nbox, const float* pres, const float* depth, int ld_pd, void* rec, int B, int HW, int I, int P, int ac,
                hipStream_t s) {
    if (ac || P != R3_P || HW > R3_MAXHW || I >= (int)R3_EMPTY || (reinterpret_cast<uintptr_t>(rec) & 15)) return SPAIR_ERR_UNSUPPORTED;
    const dim3 grid((B * HW + 255) / 256), block(256);
    if ((I & (I - 1)) == 0)
        hipLaunchKernelGGL((k_render_prep<0, 1>), grid, block, 0, s, nbox, pres, depth, ld_pd, reinterpret_cast<RenderRec*>(rec), B, HW, I, P);
    else
        hipLaunchKernelGGL((k_render_prep<0, 0>), grid, block, 0, s, nbox, pres, depth, ld_pd, reinterpret_cast<RenderRec*>(rec), B, HW, I, P);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int render_fwd_mma(const void* S16, int ld_s, const void* rec, const float* x, float* recon, float* aux, float* bce_partial, int B, int HW,
                   int I, int P, int ac, hipStream_t s) {
    if (ac || P != R3_P || ld_s != R3_P * R3_P * 2 || HW > R3_MAXHW || I >= (int)R3_EMPTY) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * R3_SPRB >= 0xfffffff0ull - 64) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S16) & 15) || (reinterpret_cast<uintptr_t>(rec) & 15)) return SPAIR_ERR_UNSUPPORTED;
    const int t = (I + RT - 1) / RT;
    const dim3 grid(B * t * t), block(256);
    const unsigned s_bytes = (unsigned)((size_t)B * HW * R3_SPRB);
    if ((I & (I - 1)) == 0)
        hipLaunchKernelGGL((k_render_fwd_mma<1>), grid, block, 0, s, S16, s_bytes, reinterpret_cast<const RenderRec*>(rec), x, recon,
                           reinterpret_cast<float2*>(aux), bce_partial, B, HW, I);
    else
        hipLaunchKernelGGL((k_render_fwd_mma<0>), grid, block, 0, s, S16, s_bytes, reinterpret_cast<const RenderRec*>(rec), x, recon,
                           reinterpret_cast<float2*>(aux), bce_partial, B, HW, I);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
