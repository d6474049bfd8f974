// K6, second generation (reference: models.py:485-547, stn(inverse=True) modules.py:256-269).
//
// FORWARD  k_render_fwd2 -- "coalesced gather with LDS-staged bilinear taps":
//   one workgroup (4 waves) per (sample, 16 x 16 pixel tile); wave w owns the 16 x 4 pixel strip w of the tile.
//   1. cull: the HW objects of the sample against the tile (one object per thread, ballot-compacted);
//   2. per pass of <= RF_TC surviving objects: their inverse-affine parameters go to LDS, every wave culls them again
//      against ITS strip and notes which sprite rows the strip can touch; the separable x taps of the 16 tile columns
//      (texel offset, two weights -- taps that fall on the zero padding already carry weight 0) are tabulated once
//      per (object, column) and shared by the four strips;
//   3. per wave, chunks of objects whose needed sprite rows fit the wave's LDS pool: the rows are CONTIGUOUS bytes of
//      the [N][P][P][2] sprite array (16-byte aligned), so they are copied HBM/L2 -> LDS by direct-to-LDS loads
//      (global_load_lds_dwordx4, no VGPR round trip, one wave-instruction per <= 9 rows); the y taps of the 4 strip
//      rows are tabulated per object; then each lane composites its pixel: two table reads, four 4-byte (bf16 grey,
//      alpha) taps from LDS, importance rebuilt per tap, ~45 VALU instructions per (pixel, object) pair
//      (the first-generation kernel issued ~150 and four scattered global loads).
//   Nothing is synchronised across waves inside step 3 (the pool and the y table are wave-private).
//
// The forward math per (pixel, object) is the same as in k_render_fwd (render.hip) up to the order of one multiplication
// (alpha * pres is applied to the interpolated alpha instead of to each tap).
#include "render_common.h"

#ifndef RF_TC
#define RF_TC 32          // objects per tile pass (<= 64: the per-strip cull is one ballot)
#endif
#ifndef RF_ROWS
#define RF_ROWS 56        // sprite rows a wave stages per chunk (>= P)
#endif
#ifndef RF_MCH
#define RF_MCH 16         // objects per chunk (y-table entries)
#endif
#ifndef RF_DMA
#define RF_DMA 1          // 1: direct-to-LDS loads; 0: through registers (A/B)
#endif

struct RfCand {
    float ax, bx, ay, by, pres, pd;
    int row, pad;
};

// One axis of the bilinear footprint of source coordinate s on a P-texel sprite with zero padding: taps i0 and i0 + 1, both inside
// [0, P-1], with weights w0 / w1 (a tap that falls on the padding has its weight moved to 0).  Returns false (weights 0) when the
// coordinate is outside (-1, P), where the padded sprite is zero.
__device__ __forceinline__ bool rf_axis(float s, int P, int& i0, float& w0, float& w1) {
    const bool cov = s > -1.f && s < (float)P;
    const float f0 = floorf(s);
    const float f = s - f0;
    int i = (int)fminf(fmaxf(f0, -1.f), (float)(P - 1));
    float a = 1.f - f, b = f;
    if (i < 0) { a = b; b = 0.f; i = 0; }                 // taps (-1, 0): the padding tap drops out
    else if (i >= P - 1) { b = a; a = 0.f; i = P - 2; }   // taps (P-1, P)
    if (!cov) { a = 0.f; b = 0.f; i = 0; }
    i0 = i; w0 = a; w1 = b;
    return cov;
}

__device__ __forceinline__ int rf_scan_incl(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

template <bool S16>
__device__ __forceinline__ float2 rf_tap(const char* p) {
    if constexpr (S16) {
        const unsigned u = *reinterpret_cast<const unsigned*>(p);
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float2*>(p);
    }
}

__host__ __device__ inline int rf_shared_bytes() { return (RF_TC * 32 + RF_TC * 256 + 512 + 16 * RF_TC + 32 + 16 + 15) & ~15; }
__host__ __device__ inline int rf_wave_bytes(int P, int texb) { return RF_MCH * 64 + RF_ROWS * P * texb; }

template <bool S16, int PT>
__global__ __launch_bounds__(256) void k_render_fwd2(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                     const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                     const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                     float* __restrict__ bce_partial, int B, int HW, int I, int Prt, int ac) {
    extern __shared__ __attribute__((aligned(16))) char sm2[];
    constexpr int TEXB = S16 ? 4 : 8;                 // bytes per (grey, alpha) texel
    constexpr int ES = S16 ? 2 : 4;                   // bytes per sprite element
    const int P = PT ? PT : Prt;
    const int ROWB = P * TEXB;
    const int POOL = RF_ROWS * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RfCand* cand = reinterpret_cast<RfCand*>(sm2);
    float4* xtab = reinterpret_cast<float4*>(sm2 + RF_TC * 32);
    unsigned short* tl = reinterpret_cast<unsigned short*>(sm2 + RF_TC * 32 + RF_TC * 256);
    unsigned* wl = reinterpret_cast<unsigned*>(tl + 256);            // [4][RF_TC]: object | first row << 8 | rows << 16
    int* cnt = reinterpret_cast<int*>(wl + 4 * RF_TC);               // [0..3] tile hits per culling wave, [4..7] hits per strip
    float* red = reinterpret_cast<float*>(cnt + 8);
    char* wbase = sm2 + rf_shared_bytes() + wave * rf_wave_bytes(P, TEXB);
    float4* ytab = reinterpret_cast<float4*>(wbase);                 // [RF_MCH][4]
    char* pool = wbase + RF_MCH * 64;
    const unsigned pool_off = (unsigned)(pool - sm2);

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
    const int lx = lane & 15, ly4 = lane >> 4;
    const int px = tx0 + lx, py = ty0 + 4 * wave + ly4;
    const bool inside = px < I && py < I;
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
    const int sy0 = ty0 + 4 * wave;                               // first pixel row of this wave's strip
    const unsigned long long below = (1ull << lane) - 1ull;
    const char* Sb = reinterpret_cast<const char*>(S);

    float num = 0.f, den = 0.f;
    for (int k0 = 0; k0 < HW; k0 += 256) {
        // ---- 1. cull 256 objects against the tile
        {
            const int k = k0 + tid;
            bool hit = false;
            if (k < HW) {
                const int r = k * B + b;
                const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
                const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
                const float ax = 1.f / nb.z, bx = -tx / nb.z, ay = 1.f / nb.w, by = -ty / nb.w;
                // the zero-padded sprite is non-zero for source coords in (-1, P)
                hit = src_of(ax, bx, tx1, I, P, ac) > -1.f && src_of(ax, bx, tx0, I, P, ac) < (float)P &&
                      src_of(ay, by, ty1, I, P, ac) > -1.f && src_of(ay, by, ty0, I, P, ac) < (float)P;
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) cnt[wave] = __popcll(bal);
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += cnt[w];
            if (hit) tl[base + __popcll(bal & below)] = (unsigned short)tid;
        }
        __syncthreads();
        const int nt = cnt[0] + cnt[1] + cnt[2] + cnt[3];
        for (int p0 = 0; p0 < nt; p0 += RF_TC) {
            const int ntc = min(RF_TC, nt - p0);
            // ---- 2a. the pass's objects: parameters (wave 0 writes them), per-strip cull and sprite row range (every wave for its strip)
            {
                bool hs = false;
                int v0 = 0, nr = 2;
                if (lane < ntc) {
                    const int kk = k0 + tl[p0 + lane];
                    const int r = kk * B + b;
                    const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
                    const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
                    const float ax = 1.f / nb.z, bx = -tx / nb.z, ay = 1.f / nb.w, by = -ty / nb.w;
                    if (wave == 0) {
                        RfCand c;
                        c.ax = ax; c.bx = bx; c.ay = ay; c.by = by;
                        c.pres = pres[(size_t)r * ld_pd];
                        c.pd = c.pres * depth[(size_t)r * ld_pd];
                        c.row = r; c.pad = 0;
                        cand[lane] = c;
                    }
                    int lo = P, hi = -1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int yy = sy0 + q;
                        if (yy < I) {
                            float gd, w0, w1;
                            int i0;
                            if (rf_axis(src_from_base(ay, by, stn_base(yy, I, ac), P, ac, gd), P, i0, w0, w1)) {
                                lo = min(lo, i0);
                                hi = max(hi, i0 + 1);
                            }
                        }
                    }
                    hs = hi >= 0;
                    v0 = lo; nr = hi - lo + 1;
                }
                const unsigned long long bs = __ballot(hs);
                if (hs) wl[wave * RF_TC + __popcll(bs & below)] = (unsigned)lane | ((unsigned)v0 << 8) | ((unsigned)nr << 16);
                if (lane == 0) cnt[4 + wave] = __popcll(bs);
            }
            __syncthreads();
            // ---- 2b. x taps of the 16 tile columns, per object
            for (int e = tid; e < ntc * 16; e += 256) {
                const RfCand cd = cand[e >> 4];
                const int xx = min(tx0 + (e & 15), I - 1);
                float gd, w0, w1;
                int i0;
                rf_axis(src_from_base(cd.ax, cd.bx, stn_base(xx, I, ac), P, ac, gd), P, i0, w0, w1);
                xtab[e] = make_float4(__uint_as_float((unsigned)(i0 * TEXB)), w0, w1, cd.pres);
            }
            __syncthreads();
            // ---- 3. this wave's strip
            const int nc = __builtin_amdgcn_readfirstlane(cnt[4 + wave]);
            unsigned ent = 0;
            int bytes = 0;
            unsigned goff = 0;
            if (lane < nc) {
                ent = wl[wave * RF_TC + lane];
                bytes = (int)((ent >> 16) & 0xffu) * ROWB;
                goff = (unsigned)cand[ent & 0xffu].row * (unsigned)(ld_s * ES) + ((ent >> 8) & 0xffu) * (unsigned)ROWB;
            }
            const int cum = rf_scan_incl(bytes, lane);
            int start = 0;
            while (start < nc) {
                const int cbase = start ? __builtin_amdgcn_readlane(cum, start - 1) : 0;
                const bool fits = lane >= start && lane < nc && (cum - cbase) <= POOL && (lane - start) < RF_MCH;
                const int m = __builtin_amdgcn_readfirstlane(__popcll(__ballot(fits)));
                // stage the chunk's sprite rows
                for (int j = 0; j < m; ++j) {
                    const int e = start + j;
                    const unsigned so = __builtin_amdgcn_readlane(goff, e);
                    const int nb_ = __builtin_amdgcn_readlane(bytes, e);
                    const int slot = __builtin_amdgcn_readlane(cum, e) - nb_ - cbase;
                    for (int o0 = 0; o0 < nb_; o0 += 1024) {
                        if (o0 + lane * 16 < nb_) {
#if RF_DMA
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Sb + (size_t)so + o0 + lane * 16),
                                                             (__attribute__((address_space(3))) void*)(pool + slot + o0), 16, 0, 0);
#else
                            *reinterpret_cast<uint4*>(pool + slot + o0 + lane * 16) = *reinterpret_cast<const uint4*>(Sb + (size_t)so + o0 + lane * 16);
#endif
                        }
                    }
                }
                // y taps of the strip's 4 pixel rows, per object of the chunk
                {
                    const int j = lane >> 2, q = lane & 3;
                    const int e = start + min(j, m - 1);
                    const unsigned en = __shfl(ent, e);
                    const int slot = __shfl(cum, e) - __shfl(bytes, e) - cbase;
                    if (j < m) {
                        const RfCand cd = cand[en & 0xffu];
                        const int yy = min(sy0 + q, I - 1);
                        float gd, w0, w1;
                        int i0;
                        const int v0 = (int)((en >> 8) & 0xffu);
                        if (!rf_axis(src_from_base(cd.ay, cd.by, stn_base(yy, I, ac), P, ac, gd), P, i0, w0, w1)) i0 = v0;
                        ytab[lane] = make_float4(__uint_as_float(pool_off + (unsigned)(slot + (i0 - v0) * ROWB)), w0, w1, cd.pd);
                    }
                }
#if RF_DMA
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                // composite
                for (int j = 0; j < m; ++j) {
                    const unsigned en = __builtin_amdgcn_readlane(ent, start + j);
                    const float4 xi = xtab[(en & 0xffu) * 16 + lx];
                    const float4 yi = ytab[j * 4 + ly4];
                    const char* tp = sm2 + (__float_as_uint(xi.x) + __float_as_uint(yi.x));
                    const float2 t00 = rf_tap<S16>(tp), t01 = rf_tap<S16>(tp + TEXB), t10 = rf_tap<S16>(tp + ROWB), t11 = rf_tap<S16>(tp + ROWB + TEXB);
                    const float w00 = yi.y * xi.y, w01 = yi.y * xi.z, w10 = yi.z * xi.y, w11 = yi.z * xi.z;
                    const float pd = yi.w;
                    float g = w00 * t00.x, a = w00 * t00.y, mm = w00 * fmaxf(t00.y * pd, 0.01f);
                    g = fmaf(w01, t01.x, g); a = fmaf(w01, t01.y, a); mm = fmaf(w01, fmaxf(t01.y * pd, 0.01f), mm);
                    g = fmaf(w10, t10.x, g); a = fmaf(w10, t10.y, a); mm = fmaf(w10, fmaxf(t10.y * pd, 0.01f), mm);
                    g = fmaf(w11, t11.x, g); a = fmaf(w11, t11.y, a); mm = fmaf(w11, fmaxf(t11.y * pd, 0.01f), mm);
                    a *= xi.w;
                    num += g * a * (mm + 1e-9f);
                    den += mm;
                }
                start += m;
            }
            __syncthreads();
        }
        if (k0 + 256 < HW) __syncthreads();      // the next cull rewrites cnt[0..3] / tl
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
            aux[pi] = make_float2(gr * invD, pre);
        }
    }
    bce = block_reduce_sum_256(bce, red);
    if (tid == 0) bce_partial[blockIdx.x] = bce;
}

// SPAIR_ERR_UNSUPPORTED: the caller falls back to the first-generation kernel
int render_fwd2(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x,
                float* recon, float* aux, float* bce_partial, int B, int HW, int I, int P, int ac, int s_bf16, hipStream_t s) {
    const int texb = s_bf16 ? 4 : 8, es = s_bf16 ? 2 : 4;
    if ((P * texb) % 16 != 0 || ((size_t)ld_s * es) % 16 != 0 || P > RF_ROWS || P < 2 || P > 255) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * ld_s * es >= (1ull << 32)) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S) & 15) != 0) return SPAIR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)rf_shared_bytes() + 4 * (size_t)rf_wave_bytes(P, texb);
    if (lds > 160 * 1024) return SPAIR_ERR_UNSUPPORTED;
    const int t = (I + RT - 1) / RT;
    const dim3 grid(B * t * t), block(256);
    float2* aux2 = reinterpret_cast<float2*>(aux);
#define RF_LAUNCH(S16_, PT_)                                                                                                            \
    do {                                                                                                                                \
        if (lds > 64 * 1024 &&                                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_fwd2<S16_, PT_>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                (int)lds) != hipSuccess)                                                                                \
            return SPAIR_ERR_LAUNCH;                                                                                                    \
        hipLaunchKernelGGL((k_render_fwd2<S16_, PT_>), grid, block, lds, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon, aux2,           \
                           bce_partial, B, HW, I, P, ac);                                                                               \
    } while (0)
    if (s_bf16) { if (P == 28) RF_LAUNCH(true, 28); else RF_LAUNCH(true, 0); }
    else { if (P == 28) RF_LAUNCH(false, 28); else RF_LAUNCH(false, 0); }
#undef RF_LAUNCH
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
