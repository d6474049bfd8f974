// K6, second generation (reference: models.py:485-547, stn(inverse=True) modules.py:256-269).
//
// FORWARD  k_render_fwd3 -- "coalesced gather with LDS-staged bilinear taps":
//   one workgroup (4 waves) per (sample, 16 x 16 pixel tile); wave w owns the 16 x 4 pixel strip w of the tile.
//   1. cull: the HW objects of the sample against the tile (one object per thread, ballot-compacted);
//   2. per pass of <= RF_TC surviving objects: their inverse-affine parameters go to LDS; the separable x / y taps of the 16 tile columns
//      and rows (texel offset, two weights -- taps that fall on the zero padding already carry weight 0) are tabulated once per
//      (object, column / row) by all 256 threads; the per-strip cull, the staged row range and the LDS slot follow from the y table;
//   3. per wave, chunks of objects whose needed sprite rows fit the wave's LDS pool: the rows are CONTIGUOUS bytes of the [N][P][P][2]
//      sprite array (16-byte aligned), copied HBM/L2 -> LDS by direct-to-LDS loads (global_load_lds_dwordx4, no VGPR round trip); then
//      each lane composites its pixel: two table reads, four 4-byte (fp16 grey, alpha) taps from LDS, importance rebuilt per tap,
//      ~45 VALU instructions per (pixel, object) pair (the first-generation kernel issued ~150 and four scattered global loads).
//   Nothing is synchronised across waves inside step 3 (the pool is wave-private).
// (k_render_fwd2, the first form of this kernel with ~930 preparation instructions per wave and tile, is retired; the band variant
//  k_render_fwd4 and the region-resident k_render_fwd5 were measured slower and live in tools/exp/.)
//
// The forward math per (pixel, object) is the same as in k_render_fwd (render.hip) up to the order of one multiplication
// (alpha * pres is applied to the interpolated alpha instead of to each tap).
#include <stdlib.h>
#include "render_common.h"

int render_prep_supported(int HW, int I, int P, int ac);                        // render3.hip
const void* render_rec_cull(const void* rec, int B, int HW);
const void* render_rec_bwd(const void* rec, int B, int HW);

#define RF_TC 32          // objects per tile pass (<= 64: the per-strip cull is one ballot)

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
        return sprite_unpack(*reinterpret_cast<const unsigned*>(p));
    } else {
        return *reinterpret_cast<const float2*>(p);
    }
}

// ---------------------------------------------------------------------------------------------
// k_render_fwd3: per-tile preparation ~450 instructions per wave (k_render_fwd2's ~930 outweighed the composite loop itself):
//   * the tile cull uses v_rcp and a small safety margin (false positives only cost zero weights); the exact inverse-affine parameters
//     (IEEE divisions, as the backward computes them) are formed once per surviving object by the first lanes of wave 0;
//   * BOTH tap tables (16 tile columns, 16 tile rows per object) are built once per tile pass by all 256 threads; the per-strip cull,
//     the staged row range and the LDS slot of every object then follow from the y table in a handful of instructions per wave, and
//     there is no per-chunk y table any more: the composite loop adds a per-object (scalar) slot base to the tabulated row index;
//   * the per-strip object list is a 64-bit ballot in scalar registers, walked with s_ff1.
// ---------------------------------------------------------------------------------------------
#define RF3_ROWS 48       // sprite rows a wave stages per chunk (>= P)
__host__ __device__ inline int rf3_shared_bytes() { return RF_TC * 32 + 2 * RF_TC * 256 + 512 + 64 + 16; }
__host__ __device__ inline int rf3_wave_bytes(int P, int texb) { return RF3_ROWS * P * texb; }

// the backward's sprite read is its last use in the step: loaded non-temporally (with the non-temporal d-logit stores 0.267 -> 0.254 ms)
__device__ __forceinline__ uint4 rb2_ld16(const char* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <bool S16, int PT, int AC, int IP2>
__global__ __launch_bounds__(256) void k_render_fwd3(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                     const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                     const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                     float* __restrict__ bce_partial, int B, int HW, int I, int Prt) {
    extern __shared__ __attribute__((aligned(16))) char sm3[];
    constexpr int TEXB = S16 ? 4 : 8;                 // bytes per (grey, alpha) texel
    constexpr int ES = S16 ? 2 : 4;                   // bytes per sprite element
    const int P = PT ? PT : Prt;
    const int ROWB = P * TEXB;
    const int POOL = RF3_ROWS * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RfCand* cand = reinterpret_cast<RfCand*>(sm3);
    float4* xtab = reinterpret_cast<float4*>(sm3 + RF_TC * 32);                    // [RF_TC][16] {tap byte offset in the row, w0, w1, pres}
    float4* ytab = xtab + RF_TC * 16;                                              // [RF_TC][16] {first tap row, w0, w1, pres * depth}
    unsigned short* tl = reinterpret_cast<unsigned short*>(ytab + RF_TC * 16);
    int* cnt = reinterpret_cast<int*>(tl + 256);
    float* red = reinterpret_cast<float*>(cnt + 16);
    char* pool = sm3 + rf3_shared_bytes() + wave * rf3_wave_bytes(P, TEXB);
    const unsigned pool_off = (unsigned)(pool - sm3);
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
    const int lx = lane & 15, ly4 = lane >> 4;
    const int px = tx0 + lx, py = ty0 + 4 * wave + ly4;
    const bool inside = px < I && py < I;
    const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
    const unsigned long long below = (1ull << lane) - 1ull;
    const char* Sb = reinterpret_cast<const char*>(S);
    // this pixel's target value: wanted only in the epilogue, fetched now
    const size_t pi = ((size_t)b * I + min(py, I - 1)) * I + min(px, I - 1);
    const float xv = x[pi];
    // base coordinates of the tile's first / last column and row (for the cull)
    float gd;
    const float bx0 = rf_base<AC, IP2>(tx0, I, inv_I), bx1 = rf_base<AC, IP2>(tx1, I, inv_I);
    const float by0 = rf_base<AC, IP2>(ty0, I, inv_I), by1 = rf_base<AC, IP2>(ty1, I, inv_I);

    float num = 0.f, den = 0.f;
    for (int k0 = 0; k0 < HW; k0 += 256) {
        // ---- 1. cull 256 objects against the tile: reciprocal instead of the IEEE divisions, bounds widened by 1e-2 texel
        {
            const int k = k0 + tid;
            bool hit = false;
            if (k < HW) {
                const int r = k * B + b;
                const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
                const float ax = __builtin_amdgcn_rcpf(nb.z), ay = __builtin_amdgcn_rcpf(nb.w);
                const float bx = -(2.f * nb.x - 1.f) * ax, by = -(2.f * nb.y - 1.f) * ay;
                const float lo = -1.01f, hi = (float)P + 0.01f;
                hit = src_from_base(ax, bx, bx1, P, AC, gd) > lo && src_from_base(ax, bx, bx0, P, AC, gd) < hi &&
                      src_from_base(ay, by, by1, P, AC, gd) > lo && src_from_base(ay, by, by0, P, AC, gd) < hi;
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
            // ---- 2. exact parameters of the pass's objects (the same expressions as the backward's)
            if (tid < ntc) {
                const int kk = k0 + tl[p0 + tid];
                const int r = kk * B + b;
                const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
                const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
                RfCand c;
                c.ax = 1.f / nb.z; c.bx = -tx / nb.z; c.ay = 1.f / nb.w; c.by = -ty / nb.w;
                c.pres = pres[(size_t)r * ld_pd];
                c.pd = c.pres * depth[(size_t)r * ld_pd];
                c.row = r; c.pad = 0;
                cand[tid] = c;
            }
            __syncthreads();
            // ---- 3. tap tables: 16 tile columns and 16 tile rows per object
            for (int e = tid; e < ntc * 32; e += 256) {
                const int c = e >> 5, yaxis = (e >> 4) & 1, idx = e & 15;
                const RfCand cd = cand[c];
                const int pos = min((yaxis ? ty0 : tx0) + idx, I - 1);
                float w0, w1;
                int i0;
                rf_axis(src_from_base(yaxis ? cd.ay : cd.ax, yaxis ? cd.by : cd.bx, rf_base<AC, IP2>(pos, I, inv_I), P, AC, gd), P, i0, w0, w1);
                if (yaxis) ytab[c * 16 + idx] = make_float4(__int_as_float(i0), w0, w1, cd.pd);
                else xtab[c * 16 + idx] = make_float4(__uint_as_float((unsigned)(i0 * TEXB)), w0, w1, cd.pres);
            }
            __syncthreads();
            // ---- 4. this wave's strip: which objects reach it, which sprite rows they need, where those rows go
            bool hs = false;
            int bytes = 0, v0 = 0;
            unsigned goff = 0;
            if (lane < ntc) {
                float4* ye = ytab + lane * 16 + 4 * wave;
                int lo = P, hi = -1;
                bool cv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 e = ye[q];
                    cv[q] = (e.y + e.z) > 0.f && (ty0 + 4 * wave + q) < I;
                    const int i0 = __float_as_int(e.x);
                    lo = cv[q] ? min(lo, i0) : lo;
                    hi = cv[q] ? max(hi, i0 + 1) : hi;
                }
                hs = hi >= 0;
                if (hs) {
                    v0 = lo;
                    bytes = (hi - lo + 1) * ROWB;
                    goff = (unsigned)cand[lane].row * (unsigned)(ld_s * ES) + (unsigned)(lo * ROWB);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (!cv[q]) reinterpret_cast<int*>(ye + q)[0] = lo;          // zero-weight rows keep their taps inside the staged rows
                }
            }
            const int cum = rf_scan_incl(bytes, lane);
            const unsigned sbase = pool_off + (unsigned)(cum - bytes) - (unsigned)(v0 * ROWB);     // + tabulated row * ROWB = first tap's row
            unsigned long long todo = __ballot(hs);
            int cbase = 0;
            while (todo) {
                const bool fits = ((todo >> lane) & 1ull) && (cum - cbase) <= POOL;
                const unsigned long long chunk = __ballot(fits);                     // a prefix of `todo`: cum is monotone
                // stage the chunk's sprite rows
                for (unsigned long long m = chunk; m; m &= m - 1) {
                    const int c = __builtin_ctzll(m);
                    const unsigned so = __builtin_amdgcn_readlane(goff, c);
                    const int nb_ = __builtin_amdgcn_readlane(bytes, c);
                    const int slot = __builtin_amdgcn_readlane(cum, c) - nb_ - cbase;
                    for (int o0 = 0; o0 < nb_; o0 += 1024) {
                        if (o0 + lane * 16 < nb_) {
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Sb + (size_t)so + o0 + lane * 16),
                                                             (__attribute__((address_space(3))) void*)(pool + slot + o0), 16, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // composite
                for (unsigned long long m = chunk; m; m &= m - 1) {
                    const int c = __builtin_ctzll(m);
                    const unsigned sb = __builtin_amdgcn_readlane(sbase, c) - (unsigned)cbase;
                    const float4 xi = xtab[c * 16 + lx];
                    const float4 yi = ytab[c * 16 + 4 * wave + ly4];
                    const char* tp = sm3 + (__float_as_uint(xi.x) + (unsigned)__float_as_int(yi.x) * (unsigned)ROWB + sb);
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
                const int last = 63 - __builtin_clzll(chunk);
                cbase = __builtin_amdgcn_readlane(cum, last);
                todo &= ~chunk;
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
    if ((P * texb) % 16 != 0 || ((size_t)ld_s * es) % 16 != 0 || P > RF3_ROWS || P < 2 || P > 255) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * ld_s * es >= (1ull << 32)) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S) & 15) != 0) return SPAIR_ERR_UNSUPPORTED;
    const int t = (I + RT - 1) / RT;
    const dim3 grid(B * t * t), block(256);
    float2* aux2 = reinterpret_cast<float2*>(aux);
    {
        const size_t lds3 = (size_t)rf3_shared_bytes() + 4 * (size_t)rf3_wave_bytes(P, texb);
        if (lds3 > 160 * 1024) return SPAIR_ERR_UNSUPPORTED;
        const bool ip2 = (I & (I - 1)) == 0;
#define RF3_LAUNCH(S16_, PT_, AC_, IP2_)                                                                                                \
    do {                                                                                                                                \
        if (lds3 > 64 * 1024 &&                                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_fwd3<S16_, PT_, AC_, IP2_>),                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess)                                   \
            return SPAIR_ERR_LAUNCH;                                                                                                    \
        hipLaunchKernelGGL((k_render_fwd3<S16_, PT_, AC_, IP2_>), grid, block, lds3, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon,     \
                           aux2, bce_partial, B, HW, I, P);                                                                             \
    } while (0)
        if (s_bf16) {
            if (P == 28 && !ac && ip2) RF3_LAUNCH(true, 28, 0, 1);
            else if (ac) RF3_LAUNCH(true, 0, 1, 0);
            else RF3_LAUNCH(true, 0, 0, 0);
        } else {
            if (P == 28 && !ac && ip2) RF3_LAUNCH(false, 28, 0, 1);
            else if (ac) RF3_LAUNCH(false, 0, 1, 0);
            else RF3_LAUNCH(false, 0, 0, 0);
        }
#undef RF3_LAUNCH
        SPAIR_CHECK_LAUNCH();
        return SPAIR_OK;
    }
}

// ---------------------------------------------------------------------------------------------
// BACKWARD  k_render_bwd2 (bf16 step: fp16 sprites in, bf16 d-logits out) -- one WAVE per object, nothing synchronised across
// waves, no atomics:
//   * the sprite goes to LDS once as a zero-bordered image of 8-byte texels {(grey, alpha) fp16 pair, importance fp32};
//   * pass A walks the object's pixel footprint in chunks of <= 16 rows x 32 columns (32 or 16 pixels x 2 or 4 rows per
//     iteration): re-samples the sprite (taps through per-column / per-row tables), forms the three per-pixel adjoints
//     (wrt interpolated grey, alpha * pres, importance), accumulates d z_where in registers and leaves the adjoints in LDS
//     as bf16 tiles [channel][pixel row][pixel column];
//   * pass B is the transpose of the bilinear sampling, which is SEPARABLE:  dS_c = Wy^T . adj_c . Wx  with the hat weights
//     Wx[px][u] = max(0, 1 - |sx(px) - u|), Wy[py][v] likewise.  Both products run on the matrix cores (fp32 accumulate):
//     T_c = adj_c . Wx per chunk (v_mfma_f32_16x16x32_bf16, K = 32 pixel columns), then dS_c += Wy^T . T_c
//     (v_mfma_f32_16x16x16_bf16, K = the chunk's 16 pixel rows) with the accumulator tile of T_c re-used directly as the B
//     operand: its rows ARE the K index in that instruction's operand layout (row 4q + j of lane group q, element j).
//     The first-generation kernel spent ~2/3 of its instructions in the per-texel gather loops this replaces.
//   * epilogue: sigmoid' and the logit scales per texel, d pres / d depth reductions, d-logits staged through LDS and written
//     as whole 16-byte pieces (the sprite gradient is contiguous).
// The adjoints and the hat weights are rounded to bf16 for the matrix products (the d-logits are stored as bf16 anyway);
// z_where gradients are fp32 throughout (pass A), pres / depth gradients come from the texel sums of the products.
// ---------------------------------------------------------------------------------------------
#define RB2_ADJ_LD 40                      // bf16 elements per adjoint tile row (80 B; 96 B would make the ds_read_b128 fragments conflict-free on
                                           // gfx950's 16-lane groups but costs a wave of occupancy: measured equal, 0.252 vs 0.252 ms; 112 B: 0.267)
#define RB2_ROWS 16                        // pixel rows per chunk
#define RB2_WAVES_PER_SIMD 4            // register budget: 128 (VGPR + AGPR)
#define RB2_BIG 1.0e9f                      // source coordinate of a padding pixel: every hat weight 0
#define RB2_ADJ_BYTES (3 * RB2_ROWS * RB2_ADJ_LD * 2)

__host__ __device__ inline int rb2_lds_bytes(int P) { return (P + 2) * (P + 2) * 8 + RB2_ADJ_BYTES + 32 * 16 + RB2_ROWS * 16 + 32 * 4 + RB2_ROWS * 4; }

__device__ __forceinline__ float rb2_hat(float s, float c) { return fmaxf(1.f - fabsf(s - c), 0.f); }
// 8 hat weights max(0, 1 - |s_j - c|) as a bf16 MFMA fragment
__device__ __forceinline__ bf16x8 rb2_hat8(const float4 s0, const float4 s1, float c) {
    bf16x8 w;
    w[0] = (__bf16)rb2_hat(s0.x, c); w[1] = (__bf16)rb2_hat(s0.y, c); w[2] = (__bf16)rb2_hat(s0.z, c); w[3] = (__bf16)rb2_hat(s0.w, c);
    w[4] = (__bf16)rb2_hat(s1.x, c); w[5] = (__bf16)rb2_hat(s1.y, c); w[6] = (__bf16)rb2_hat(s1.z, c); w[7] = (__bf16)rb2_hat(s1.w, c);
    return w;
}
__device__ __forceinline__ bf16x4 rb2_hat4(const float4 s, float c) {
    bf16x4 w;
    w[0] = (__bf16)rb2_hat(s.x, c); w[1] = (__bf16)rb2_hat(s.y, c); w[2] = (__bf16)rb2_hat(s.z, c); w[3] = (__bf16)rb2_hat(s.w, c);
    return w;
}
typedef short rb2_s16x4 __attribute__((ext_vector_type(4)));

struct Rb2Taps { uint2 q00, q01, q10, q11; };
__device__ __forceinline__ Rb2Taps rb2_ld_taps(const char* tp, int rowb) {
    Rb2Taps t;
    t.q00 = *reinterpret_cast<const uint2*>(tp); t.q01 = *reinterpret_cast<const uint2*>(tp + 8);
    t.q10 = *reinterpret_cast<const uint2*>(tp + rowb); t.q11 = *reinterpret_cast<const uint2*>(tp + rowb + 8);
    return t;
}

// REC: the inverse-affine parameters and the pixel footprint come from the forward's per-object records (render3.hip: k_render_prep wrote
// them with the very expressions below) instead of four IEEE divisions and two footprint searches per object
template <int PT, int AC, int IP2, bool REC>
__global__ __launch_bounds__(64, RB2_WAVES_PER_SIMD) void k_render_bwd2(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                    const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                    const float2* __restrict__ aux, const float* __restrict__ gloss,
                                                    __bf16* __restrict__ dlogits, float* __restrict__ dnbox, float* __restrict__ dpres,
                                                    float* __restrict__ ddepth, int ld_g, int B, int HW, int I, int Prt,
                                                    float obj_scale, float alpha_scale, const uint4* __restrict__ crec,
                                                    const float4* __restrict__ brec) {
    extern __shared__ __attribute__((aligned(16))) char smb[];
    const int P = PT ? PT : Prt;
    const int PS = P + 2;
    const int lane = threadIdx.x;
    uint2* Ssh = reinterpret_cast<uint2*>(smb);                                   // [PS][PS] {(grey, alpha) fp16 pair, importance}
    char* adjT = smb + PS * PS * 8;                                               // [3][RB2_ROWS][RB2_ADJ_LD] bf16
    float4* xt = reinterpret_cast<float4*>(adjT + RB2_ADJ_BYTES);                 // [32] {tap byte offset, frac, grid coord, -}
    float4* yt = xt + 32;                                                         // [RB2_ROWS]
    float* sxs = reinterpret_cast<float*>(yt + RB2_ROWS);                         // [32] source x of the chunk's pixel columns
    float* sys = sxs + 32;                                                        // [RB2_ROWS]
    // (sample, object): consecutive workgroup ids walk the samples, so with B % 8 == 0 every object of sample b lands on XCD b % 8
    const int b = blockIdx.x, k = blockIdx.y;
    const int r = k * B + b;
    // the sprite's first 256 16-byte pieces (all of it at P = 28) are requested before anything else: their latency overlaps the
    // latency of the parameter loads and the geometry arithmetic instead of following it
    const char* Sr = reinterpret_cast<const char*>(S) + (size_t)r * ld_s * 2;
    const int ppr = P >> 2, npieces = P * ppr;                      // 16-byte pieces (4 texels) per row / per sprite
    uint4 q_first[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q_first[i] = rb2_ld16(Sr + (size_t)min(64 * i + lane, npieces - 1) * 16);
    const float pr = pres[(size_t)r * ld_pd], dp = depth[(size_t)r * ld_pd], pd = pr * dp;
    const float gl = *gloss;
    const float2* auxb = aux + (size_t)b * I * I;
    float ax, bx, ay, by;
    int PX0, PX1, PY0, PY1;
    const float inv_I = 1.f / (float)I;
    const float mult = AC ? 0.5f * (float)(P - 1) : 0.5f * (float)P;
    if constexpr (REC) {
        const float4 pb = brec[(size_t)b * HW + k];
        const uint4 pc = crec[(size_t)b * HW + k];
        ax = pb.x; bx = pb.y; ay = pb.z; by = pb.w;
        PX0 = pc.z & 0xffff; PX1 = pc.z >> 16; PY0 = pc.w & 0xffff; PY1 = pc.w >> 16;     // an empty footprint has first > last
    } else {
        const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
        const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
        ax = 1.f / nb.z; bx = -tx / nb.z; ay = 1.f / nb.w; by = -ty / nb.w;
        float sx0, sxa, sy0, sya;
        src_affine(ax, bx, I, P, AC, sx0, sxa);
        src_affine(ay, by, I, P, AC, sy0, sya);
        rb2_range<AC, IP2>(ax, bx, sx0, __builtin_amdgcn_rcpf(sxa), I, inv_I, P, PX0, PX1);
        rb2_range<AC, IP2>(ay, by, sy0, __builtin_amdgcn_rcpf(sya), I, inv_I, P, PY0, PY1);
    }
    PX0 = __builtin_amdgcn_readfirstlane(PX0); PX1 = __builtin_amdgcn_readfirstlane(PX1);
    PY0 = __builtin_amdgcn_readfirstlane(PY0); PY1 = __builtin_amdgcn_readfirstlane(PY1);
    // ---- sprite -> LDS (zero border), adjoint tiles zeroed once (later chunks leave finite values under zero weights)
    {
        for (int p0 = 0; p0 < npieces; p0 += 256) {
            uint4 q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                q[i] = p0 == 0 ? q_first[i] : rb2_ld16(Sr + (size_t)min(p0 + 64 * i + lane, npieces - 1) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = p0 + 64 * i + lane;
                if (p < npieces) {
                    const int v = p / ppr, u0 = (p - v * ppr) * 4;
                    uint2* d = Ssh + (v + 1) * PS + u0 + 1;
                    const unsigned ga[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        d[t] = make_uint2(ga[t], __float_as_uint(fmaxf(sprite_unpack(ga[t]).y * pd, 0.01f)));
                }
            }
        }
        for (int e = lane; e < PS; e += 64) {
            Ssh[e] = make_uint2(0u, 0u);
            Ssh[(PS - 1) * PS + e] = make_uint2(0u, 0u);
            Ssh[e * PS] = make_uint2(0u, 0u);
            Ssh[e * PS + PS - 1] = make_uint2(0u, 0u);
        }
        for (int e = lane; e < RB2_ADJ_BYTES / 16; e += 64) reinterpret_cast<uint4*>(adjT)[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    // One wave per workgroup: every LDS hand-over below is between lanes of this wave.  wave_lds_fence() (common.h) states the order
    // of each write phase and the reads that follow it for the compiler; it emits nothing.
    wave_lds_fence();
    const int pw = PX1 - PX0 + 1;
    const int wsh = pw <= 16 ? 4 : 5, W = 1 << wsh, RPI = 64 >> wsh;
    const int col = lane & (W - 1), rsub = lane >> wsh;
    const int fr = lane & 15, fq = lane >> 4;
    const int rowb = PS * 8;
    float g_tx = 0.f, g_ty = 0.f, g_xs = 0.f, g_ys = 0.f;
    f32x4 dS[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) dS[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int py0 = PY0; py0 <= PY1; py0 += RB2_ROWS) {
        const int nrw = min(RB2_ROWS, PY1 - py0 + 1);
        if (lane < RB2_ROWS) {
            const int yy = py0 + lane;
            float gn = 0.f, s = RB2_BIG, f = 0.f;
            int off = 0;
            if (yy <= PY1) {
                s = src_from_base(ay, by, rf_base<AC, IP2>(yy, I, inv_I), P, AC, gn);
                const float f0 = fminf(fmaxf(floorf(s), -1.f), (float)(P - 1));
                f = s - f0;
                off = ((int)f0 + 1) * rowb;
            }
            yt[lane] = make_float4(__int_as_float(off), f, gn, 0.f);
            sys[lane] = s;
        }
        wave_lds_fence();
        for (int px0 = PX0; px0 <= PX1; px0 += W) {
            if (lane < 32) {
                const int xx = px0 + lane;
                float gn = 0.f, s = RB2_BIG, f = 0.f;
                int off = 0;
                if (lane < W && xx <= PX1) {
                    s = src_from_base(ax, bx, rf_base<AC, IP2>(xx, I, inv_I), P, AC, gn);
                    const float f0 = fminf(fmaxf(floorf(s), -1.f), (float)(P - 1));
                    f = s - f0;
                    off = ((int)f0 + 1) * 8;
                }
                xt[lane] = make_float4(__int_as_float(off), f, gn, 0.f);
                sxs[lane] = s;
            }
            wave_lds_fence();
            // ---- pass A: adjoints of the chunk's pixels, two row groups per trip (A / B register sets: the loads of one are issued
            // before the arithmetic of the other)
            const int nit = (nrw + RPI - 1) >> (6 - wsh);
            const float4 xe = xt[col];
            const int xx = min(px0 + col, PX1);
            const bool xvalid = px0 + col <= PX1;
            const unsigned xb = (unsigned)__float_as_int(xe.x);
            const float fx = xe.y;
            const float2* ap = auxb + xx;
            const unsigned adj0 = (unsigned)(adjT - smb) + (unsigned)(rsub * RB2_ADJ_LD + col) * 2u;
            const unsigned yt0 = (unsigned)(reinterpret_cast<char*>(yt) - smb) + (unsigned)rsub * 16u;
            struct PxIn { float2 av; float4 ye; Rb2Taps t; };
            auto load_px = [&](int it) {
                PxIn in;
                const int row = min(it * RPI + rsub, RB2_ROWS - 1);
                in.av = ap[(unsigned)(min(py0 + row, PY1) * I)];
                in.ye = *reinterpret_cast<const float4*>(smb + yt0 + (unsigned)(min(it * RPI, RB2_ROWS - RPI) * 16));
                in.t = rb2_ld_taps(smb + (xb + (unsigned)__float_as_int(in.ye.x)), rowb);
                return in;
            };
            auto do_px = [&](const PxIn& in, int it) {
                const int row = it * RPI + rsub;
                const bool valid = xvalid && (py0 + row <= PY1);
                const Rb2Taps& t = in.t;
                const float fy = in.ye.y;
                const float2 u00 = sprite_unpack(t.q00.x), u01 = sprite_unpack(t.q01.x), u10 = sprite_unpack(t.q10.x), u11 = sprite_unpack(t.q11.x);
                const float g00 = u00.x, a00 = u00.y, m00 = __uint_as_float(t.q00.y);
                const float g01 = u01.x, a01 = u01.y, m01 = __uint_as_float(t.q01.y);
                const float g10 = u10.x, a10 = u10.y, m10 = __uint_as_float(t.q10.y);
                const float g11 = u11.x, a11 = u11.y, m11 = __uint_as_float(t.q11.y);
                // separable bilinear: x first, then y; x-derivative = (right - left), y-derivative = (bottom - top)
                const float dTg = g01 - g00, dBg = g11 - g10, dTa = a01 - a00, dBa = a11 - a10, dTm = m01 - m00, dBm = m11 - m10;
                const float hTg = fmaf(fx, dTg, g00), hBg = fmaf(fx, dBg, g10), hTa = fmaf(fx, dTa, a00), hBa = fmaf(fx, dBa, a10);
                const float hTm = fmaf(fx, dTm, m00), hBm = fmaf(fx, dBm, m10);
                const float eg = hBg - hTg, ea = hBa - hTa, em = hBm - hTm;
                const float g = fmaf(fy, eg, hTg), a = fmaf(fy, ea, hTa) * pr, m = fmaf(fy, em, hTm);
                const float dgx = fmaf(fy, dBg - dTg, dTg), dax = fmaf(fy, dBa - dTa, dTa), dmx = fmaf(fy, dBm - dTm, dTm);
                const float go = valid ? in.av.x * gl : 0.f;          // dBCE/dpre / D
                const float tt = go * (m + 1e-9f);
                const float d_g = tt * a, d_a = tt * g;                // wrt grey, wrt (alpha * pres)
                const float d_m = go * (a * g - in.av.y);
                const float dap = d_a * pr;
                const float g_sx = d_g * dgx + dap * dax + d_m * dmx;  // d / d(source x), pixel units
                const float g_sy = d_g * eg + dap * ea + d_m * em;
                g_tx += g_sx; g_xs = fmaf(g_sx, xe.z, g_xs);           // scaled by cgx / cgy after the loops
                g_ty += g_sy; g_ys = fmaf(g_sy, in.ye.z, g_ys);
                char* q = smb + adj0 + (unsigned)(it * RPI * RB2_ADJ_LD * 2);      // (pairing columns into 4-byte stores by DPP was measured: 263 -> 325 us)
                *reinterpret_cast<__bf16*>(q) = (__bf16)d_g;
                *reinterpret_cast<__bf16*>(q + RB2_ROWS * RB2_ADJ_LD * 2) = (__bf16)d_a;
                *reinterpret_cast<__bf16*>(q + 2 * RB2_ROWS * RB2_ADJ_LD * 2) = (__bf16)d_m;
            };
            PxIn inA = load_px(0);
            for (int it = 0; it < nit; it += 2) {
                const PxIn inB = load_px(it + 1);
                do_px(inA, it);
                inA = load_px(it + 2);
                do_px(inB, it + 1);
            }
            wave_lds_fence();          // pass A's adjoint tiles -> pass B's fragment reads
            // ---- pass B.  First product: T_c[py][u] = sum_px adj_c[py][px] * Wx[px][u] for this chunk (fresh accumulators: nothing of
            // pass B is live during pass A).  Second product right behind it: dS_c[v][u] += sum_py Wy[py][v] * T_c[py][u].  T's
            // accumulator tile is the B operand of v_mfma_f32_16x16x16_bf16 as it stands (lane group q holds K = 4q .. 4q+3 = the tile
            // rows of its four registers); Wy^T is built on the fly in the same order.
            const float4 s0 = *reinterpret_cast<const float4*>(sxs + 8 * fq), s1 = *reinterpret_cast<const float4*>(sxs + 8 * fq + 4);
            const bf16x8 wx0 = rb2_hat8(s0, s1, (float)fr), wx1 = rb2_hat8(s0, s1, (float)(16 + fr));
            const float4 sq = *reinterpret_cast<const float4*>(sys + 4 * fq);
            const bf16x4 wy0 = rb2_hat4(sq, (float)fr), wy1 = rb2_hat4(sq, (float)(16 + fr));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(adjT + c * (RB2_ROWS * RB2_ADJ_LD * 2) + fr * (RB2_ADJ_LD * 2) + fq * 16);
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 T0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wx0, z, 0, 0, 0);
                const f32x4 T1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wx1, z, 0, 0, 0);
                bf16x4 tb0, tb1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { tb0[i] = (__bf16)T0[i]; tb1[i] = (__bf16)T1[i]; }
                dS[c][0][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(rb2_s16x4, wy0), __builtin_bit_cast(rb2_s16x4, tb0), dS[c][0][0], 0, 0, 0);
                dS[c][1][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(rb2_s16x4, wy1), __builtin_bit_cast(rb2_s16x4, tb0), dS[c][1][0], 0, 0, 0);
                dS[c][0][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(rb2_s16x4, wy0), __builtin_bit_cast(rb2_s16x4, tb1), dS[c][0][1], 0, 0, 0);
                dS[c][1][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(rb2_s16x4, wy1), __builtin_bit_cast(rb2_s16x4, tb1), dS[c][1][1], 0, 0, 0);
            }
            wave_lds_fence();          // pass B's reads of the tiles / tables -> the next chunk's writes
        }
    }
    // ---- epilogue: per texel sigmoid' and logit scales (models.py:485-492), d pres / d depth; d-logits staged in LDS (over the adjoint tiles)
    float g_pr = 0.f, g_s2a = 0.f;
    unsigned* ost = reinterpret_cast<unsigned*>(adjT);
#pragma unroll
    for (int vt = 0; vt < 2; ++vt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 16 * vt + 4 * fq + i, u = 16 * nt + fr;
                if (v < P && u < P) {
                    const uint2 sv = Ssh[(v + 1) * PS + u + 1];
                    const float2 sga = sprite_unpack(sv.x);
                    const float sg = sga.x, sa = sga.y;
                    const float s0_ = dS[0][vt][nt][i], s1_ = dS[1][vt][nt][i], s2_ = dS[2][vt][nt][i];
                    const bool act = (sa * pd) >= 0.01f;                 // importance not clamped
                    const float s2a = act ? s2_ * sa : 0.f;
                    g_pr = fmaf(s1_, sa, g_pr);
                    g_s2a += s2a;
                    const float ox = s0_ * sg * (1.f - sg) * obj_scale;
                    const float oy = (s1_ * pr + (act ? s2_ * pd : 0.f)) * sa * (1.f - sa) * alpha_scale;
                    __bf16 ob[2] = {(__bf16)ox, (__bf16)oy};
                    ost[v * P + u] = *reinterpret_cast<unsigned*>(ob);
                }
            }
    wave_lds_fence();                  // per-texel d-logit staging -> the 16-byte pieces read back below
    {
        char* dst = reinterpret_cast<char*>(dlogits + (size_t)r * ld_g);
        const int npieces = (P * P * 4) >> 4;
        for (int p = lane; p < npieces; p += 64) {
            const u32x4_t o = reinterpret_cast<const u32x4_t*>(ost)[p];
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(dst + (size_t)p * 16));
        }
    }
    const float cgx = -mult * ax, cgy = -mult * ay;                  // d(source coord)/d(t) incl. the unnormalisation
    g_tx = wave_reduce_sum(g_tx) * cgx; g_ty = wave_reduce_sum(g_ty) * cgy;
    g_xs = wave_reduce_sum(g_xs) * cgx; g_ys = wave_reduce_sum(g_ys) * cgy;
    g_pr = wave_reduce_sum(g_pr); g_s2a = wave_reduce_sum(g_s2a);
    if (lane == 0) {
        *reinterpret_cast<float4*>(dnbox + (size_t)r * 4) = make_float4(2.f * g_tx, 2.f * g_ty, g_xs, g_ys);
        dpres[r] = g_pr + g_s2a * dp;
        ddepth[r] = g_s2a * pr;
    }
}

// bf16 sprites in, bf16 d-logits out.  SPAIR_ERR_UNSUPPORTED: the caller falls back to the first-generation kernel.
int render_bwd2(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* aux,
                const float* gloss, float* dlogits, float* dnbox, float* dpres, float* ddepth, int ld_g, int B, int HW, int I, int P,
                int ac, float obj_scale, float alpha_scale, const void* rec, hipStream_t s) {
    if ((P & 3) || P > 32 || P < 4 || (ld_s & 7) || (ld_g & 7) || HW > 65535) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S) & 15) || (reinterpret_cast<uintptr_t>(dlogits) & 15)) return SPAIR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)rb2_lds_bytes(P);
    if (lds > 64 * 1024 || (size_t)P * P * 4 > RB2_ADJ_BYTES) return SPAIR_ERR_UNSUPPORTED;
    const dim3 grid(B, HW), block(64);
    const bool use_rec = rec && render_prep_supported(HW, I, P, ac);
    const uint4* crec = use_rec ? reinterpret_cast<const uint4*>(render_rec_cull(rec, B, HW)) : nullptr;
    const float4* brec = use_rec ? reinterpret_cast<const float4*>(render_rec_bwd(rec, B, HW)) : nullptr;
#define RB2_LAUNCH(PT_, AC_, IP2_, REC_)                                                                                                  \
    hipLaunchKernelGGL((k_render_bwd2<PT_, AC_, IP2_, REC_>), grid, block, lds, s, S, ld_s, nbox, pres, depth, ld_pd,                      \
                       reinterpret_cast<const float2*>(aux), gloss, reinterpret_cast<__bf16*>(dlogits), dnbox, dpres, ddepth, ld_g, B, HW, I, \
                       P, obj_scale, alpha_scale, crec, brec)
    const bool ip2 = (I & (I - 1)) == 0;
    if (P == 28 && !ac && ip2) { if (use_rec) RB2_LAUNCH(28, 0, 1, true); else RB2_LAUNCH(28, 0, 1, false); }
    else if (ac) RB2_LAUNCH(0, 1, 0, false);
    else { if (use_rec) RB2_LAUNCH(0, 0, 0, true); else RB2_LAUNCH(0, 0, 0, false); }
#undef RB2_LAUNCH
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
