// MEASURED AND REJECTED (round 4) -- kept here as the record of the experiment, not built into libspair_hip.so.
// configs[1], forward passes from the seed-3 weights (tools/exp/render_ablate.py), k_render_fwd3 = 0.2156 ms:
//   regions of 4 x 4 tiles + pipelined strips 0.3075 | regions, unpipelined 0.2849 | 2 x 2 tiles, pipelined 0.2428
//   one tile per workgroup, pipelined 0.2381 | one tile per workgroup, unpipelined (= fwd3 with the parameters formed by all 256 threads) 0.2269
// configs[3] (256^2): 0.488 ms against 0.291.  Results are bit-identical to k_render_fwd3 (every renderer / engine test passes with it).
// Why it loses: 1024 persistent workgroups are exactly one round of 4 per CU, so the slowest CU sets the time, where 16,384 one-tile
// workgroups are balanced by the dispatcher; forming every object's exact parameters (four IEEE divisions) in all 256 threads costs more
// than the second global round trip it removes; half-pool chunks double the number of DMA waits of a strip, and a strip rarely has enough
// compositing per chunk (4 objects) to cover the next chunk's round trip.
// To try it: copy to spair_pytorch_amd/csrc/, declare render_fwd3() in render.hip and call it ahead of render_fwd2().
//
// K6 forward, third generation (reference: models.py:485-547, stn(inverse=True) modules.py:256-269): k_render_fwd6.
//
// Same per-(pixel, object) arithmetic, tap tables, wave-private sprite pools and direct-to-LDS staging as k_render_fwd3 (render2.hip) -- the
// results are bit-identical -- around a different skeleton.  Round-4 ablations of k_render_fwd3 at BASELINE configs[1] (forward passes from the
// seed-3 weights, tools/exp/render_ablate.py): 0.216 ms = 0.080 (neither staging nor compositing: cull, parameters, tap tables, barriers,
// epilogue) + 0.073 (staging) + 0.077 (compositing) -- three ADDITIVE thirds: the four workgroups of a CU run the same tile program in
// lockstep, so their DMA phases and their compositing phases coincide instead of overlapping, and each tile starts with two dependent
// global round trips (nbox for the cull, then nbox / pres / depth again for the survivors' exact parameters).
//   * PERSISTENT REGIONS: a workgroup owns a region of RG x RG tiles of one sample and walks them.  The objects' exact inverse-affine
//     parameters are formed ONCE per workgroup and stay in REGISTERS (thread k holds objects k, k + 256, ...): the per-tile cull is pure
//     VALU work on registers, the survivors write their parameters straight into the pass table -- no global load is left on a tile's
//     critical path except the sprite rows themselves (the target pixel is fetched a tile ahead).
//   * PIPELINED STRIPS: a wave's pool is two halves; the rows of chunk i + 1 are requested (direct-to-LDS) right after chunk i has landed
//     and BEFORE chunk i is composited, so a wave always has sprite rows in flight while it computes (a strip with an object that does not
//     fit half a pool -- objects a few pixels tall -- takes the unpipelined loop).
#include <stdlib.h>
#include "render_common.h"

namespace {

constexpr int R6_TC = 32;          // objects per tile pass (<= 64: the per-strip cull is one ballot)
constexpr int R6_ROWS = 48;        // sprite rows in a wave's pool (>= P + 1)
#ifndef R6_RG_V
#define R6_RG_V 4
#endif
constexpr int R6_RG = R6_RG_V;           // region side in tiles

struct R6Cand {
    float ax, bx, ay, by, pres, pd;
    int row, pad;
};

__device__ __forceinline__ bool r6_axis(float s, int P, int& i0, float& w0, float& w1) {       // = rf_axis (render2.hip)
    const bool cov = s > -1.f && s < (float)P;
    const float f0 = floorf(s);
    const float f = s - f0;
    int i = (int)fminf(fmaxf(f0, -1.f), (float)(P - 1));
    float a = 1.f - f, b = f;
    if (i < 0) { a = b; b = 0.f; i = 0; }
    else if (i >= P - 1) { b = a; a = 0.f; i = P - 2; }
    if (!cov) { a = 0.f; b = 0.f; i = 0; }
    i0 = i; w0 = a; w1 = b;
    return cov;
}
template <int AC, int IP2>
__device__ __forceinline__ float r6_base(int j, int n, float inv_n) {
    if constexpr (IP2 && !AC) return (2.f * (float)j + 1.f) * inv_n - 1.f;
    else return stn_base(j, n, AC);
}
__device__ __forceinline__ int r6_scan_incl(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
template <bool S16>
__device__ __forceinline__ float2 r6_tap(const char* p) {
    if constexpr (S16) return sprite_unpack(*reinterpret_cast<const unsigned*>(p));
    else return *reinterpret_cast<const float2*>(p);
}

__host__ __device__ inline int r6_shared_bytes() { return R6_TC * 32 + 2 * R6_TC * 256 + 512 + 4 * 4 * 16 + 64; }
__host__ __device__ inline int r6_wave_bytes(int P, int texb) { return R6_ROWS * P * texb; }

template <bool S16, int PT, int AC, int IP2, int NR>
__global__ __launch_bounds__(256) void k_render_fwd6(const float* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                     const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                     const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                     float* __restrict__ bce_partial, int B, int HW, int I, int Prt) {
    extern __shared__ __attribute__((aligned(16))) char sm6[];
    constexpr int TEXB = S16 ? 4 : 8;                 // bytes per (grey, alpha) texel
    constexpr int ES = S16 ? 2 : 4;                   // bytes per sprite element
    const int P = PT ? PT : Prt;
    const int ROWB = P * TEXB;
    const int POOL = R6_ROWS * ROWB, HALF = (POOL / 2) & ~15;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    R6Cand* cand = reinterpret_cast<R6Cand*>(sm6);
    float4* xtab = reinterpret_cast<float4*>(sm6 + R6_TC * 32);                    // [R6_TC][16] {tap byte offset in the row, w0, w1, pres}
    float4* ytab = xtab + R6_TC * 16;                                              // [R6_TC][16] {first tap row, w0, w1, pres * depth}
    int* cnt = reinterpret_cast<int*>(ytab + R6_TC * 16);                          // [NR <= 4][4 waves] hit counts (+ padding)
    float* red = reinterpret_cast<float*>(cnt + 128);
    char* pool = sm6 + r6_shared_bytes() + wave * r6_wave_bytes(P, TEXB);
    const unsigned pool_off = (unsigned)(pool - sm6);
    const float inv_I = 1.f / (float)I;

    const int tiles_x = (I + RT - 1) / RT, regs_x = (tiles_x + R6_RG - 1) / R6_RG, regs = regs_x * regs_x;
    int b, reg;
    if ((B & 7) == 0) {   // XCD-aware: blocks id, id+8, ... share an XCD (round-robin dispatch): a sample's regions stay on one XCD
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = (j / regs) * 8 + xcd;
        reg = j % regs;
    } else {
        b = blockIdx.x / regs;
        reg = blockIdx.x % regs;
    }
    const int rtx0 = (reg % regs_x) * R6_RG, rty0 = (reg / regs_x) * R6_RG;
    const int ntx = min(R6_RG, tiles_x - rtx0), nty = min(R6_RG, tiles_x - rty0);
    const int lx = lane & 15, ly4 = lane >> 4;
    const unsigned long long below = (1ull << lane) - 1ull;
    const char* Sb = reinterpret_cast<const char*>(S);
    float gd;

    // ---- the sample's objects k = tid + 256 r: exact parameters (the same expressions as the backward's), once, into registers
    R6Cand my[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int k = tid + 256 * r;
        const int rowi = min(k, HW - 1) * B + b;
        const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)rowi * 4);
        const float pr = pres[(size_t)rowi * ld_pd], dp = depth[(size_t)rowi * ld_pd];
        const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
        my[r].ax = 1.f / nb.z; my[r].bx = -tx / nb.z; my[r].ay = 1.f / nb.w; my[r].by = -ty / nb.w;
        my[r].pres = pr; my[r].pd = pr * dp;
        my[r].row = k < HW ? rowi : -1; my[r].pad = 0;
    }

    const int ntiles = ntx * nty;
    auto tile_xy = [&](int ti, int& tx0, int& ty0) { tx0 = (rtx0 + ti % ntx) * RT; ty0 = (rty0 + ti / ntx) * RT; };
    // target pixel of the first tile (every later one is fetched a tile ahead)
    float xv_next;
    {
        int tx0, ty0;
        tile_xy(0, tx0, ty0);
        xv_next = x[((size_t)b * I + min(ty0 + 4 * wave + ly4, I - 1)) * I + min(tx0 + lx, I - 1)];
    }

    for (int ti = 0; ti < ntiles; ++ti) {
        int tx0, ty0;
        tile_xy(ti, tx0, ty0);
        const int px = tx0 + lx, py = ty0 + 4 * wave + ly4;
        const bool inside = px < I && py < I;
        const int tx1 = min(tx0 + RT, I) - 1, ty1 = min(ty0 + RT, I) - 1;
        const size_t pi = ((size_t)b * I + min(py, I - 1)) * I + min(px, I - 1);
        const float xv = xv_next;
        if (ti + 1 < ntiles) {
            int nx0, ny0;
            tile_xy(ti + 1, nx0, ny0);
            xv_next = x[((size_t)b * I + min(ny0 + 4 * wave + ly4, I - 1)) * I + min(nx0 + lx, I - 1)];
        }
        const float bx0 = r6_base<AC, IP2>(tx0, I, inv_I), bx1 = r6_base<AC, IP2>(tx1, I, inv_I);
        const float by0 = r6_base<AC, IP2>(ty0, I, inv_I), by1 = r6_base<AC, IP2>(ty1, I, inv_I);

        // ---- 1. cull this thread's objects against the tile (registers only); one ballot per round, then ONE barrier
        bool hit[NR];
        int pre[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float lo = -1.f, hi = (float)P;
            hit[r] = my[r].row >= 0 && src_from_base(my[r].ax, my[r].bx, bx1, P, AC, gd) > lo && src_from_base(my[r].ax, my[r].bx, bx0, P, AC, gd) < hi &&
                     src_from_base(my[r].ay, my[r].by, by1, P, AC, gd) > lo && src_from_base(my[r].ay, my[r].by, by0, P, AC, gd) < hi;
            const unsigned long long bal = __ballot(hit[r]);
            pre[r] = __popcll(bal & below);
            if (lane == 0) cnt[r * 4 + wave] = __popcll(bal);
        }
        __syncthreads();
        int slot[NR], nt = 0;
        {
            // list order = (round, wave, lane): the order k_render_fwd3 composites in (it walks the rounds one after the other)
            int c16[NR * 4];
#pragma unroll
            for (int q = 0; q < NR * 4; ++q) c16[q] = cnt[q];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int base = nt;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w < wave) base += c16[r * 4 + w];
                    nt += c16[r * 4 + w];
                }
                slot[r] = base + pre[r];
            }
        }

        float num = 0.f, den = 0.f;
        for (int p0 = 0; p0 < nt; p0 += R6_TC) {
            const int ntc = min(R6_TC, nt - p0);
            // ---- 2. the pass's objects: parameters from the owners' registers
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (hit[r] && slot[r] >= p0 && slot[r] < p0 + R6_TC) cand[slot[r] - p0] = my[r];
            __syncthreads();
            // ---- 3. tap tables: 16 tile columns and 16 tile rows per object
            for (int e = tid; e < ntc * 32; e += 256) {
                const int c = e >> 5, yaxis = (e >> 4) & 1, idx = e & 15;
                const R6Cand cd = cand[c];
                const int pos = min((yaxis ? ty0 : tx0) + idx, I - 1);
                float w0, w1;
                int i0;
                r6_axis(src_from_base(yaxis ? cd.ay : cd.ax, yaxis ? cd.by : cd.bx, r6_base<AC, IP2>(pos, I, inv_I), P, AC, gd), P, i0, w0, w1);
                if (yaxis) ytab[c * 16 + idx] = make_float4(__int_as_float(i0), w0, w1, cd.pd);
                else xtab[c * 16 + idx] = make_float4(__uint_as_float((unsigned)(i0 * TEXB)), w0, w1, cd.pres);
            }
            __syncthreads();
            // ---- 4. this wave's strip: which objects reach it, which sprite rows they need
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
            const int cum = r6_scan_incl(bytes, lane);
            // (the slot of an object inside its chunk's buffer is cum - bytes - <chunk base>; the tabulated row index is relative to the sprite)
            const int rel = (cum - bytes) - v0 * ROWB;
            unsigned long long todo = __ballot(hs);

            auto stage = [&](unsigned long long chunk, int cbase, int buf_off) {
                for (unsigned long long m = chunk; m; m &= m - 1) {
                    const int c = __builtin_ctzll(m);
                    const unsigned so = __builtin_amdgcn_readlane(goff, c);
                    const int nb_ = __builtin_amdgcn_readlane(bytes, c);
                    const int slot_ = __builtin_amdgcn_readlane(cum, c) - nb_ - cbase + buf_off;
                    for (int o0 = 0; o0 < nb_; o0 += 1024) {
                        if (o0 + lane * 16 < nb_) {
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Sb + (size_t)so + o0 + lane * 16),
                                                             (__attribute__((address_space(3))) void*)(pool + slot_ + o0), 16, 0, 0);
                        }
                    }
                }
            };
            auto composite = [&](unsigned long long chunk, int cbase, int buf_off) {
                for (unsigned long long m = chunk; m; m &= m - 1) {
                    const int c = __builtin_ctzll(m);
                    const unsigned sb = pool_off + (unsigned)(__builtin_amdgcn_readlane(rel, c) - cbase + buf_off);
                    const float4 xi = xtab[c * 16 + lx];
                    const float4 yi = ytab[c * 16 + 4 * wave + ly4];
                    const char* tp = sm6 + (__float_as_uint(xi.x) + (unsigned)__float_as_int(yi.x) * (unsigned)ROWB + sb);
                    const float2 t00 = r6_tap<S16>(tp), t01 = r6_tap<S16>(tp + TEXB), t10 = r6_tap<S16>(tp + ROWB), t11 = r6_tap<S16>(tp + ROWB + TEXB);
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
            };
            // next chunk: the longest prefix of `todo` that fits `cap` bytes (cum is monotone over the lanes)
            auto next_chunk = [&](unsigned long long td, int cbase, int cap) {
                const bool fits = ((td >> lane) & 1ull) && (cum - cbase) <= cap;
                return __ballot(fits);
            };

            #ifdef R6_NOPIPE
            const bool big = true;
#else
            const bool big = __ballot(hs && bytes > HALF) != 0ull;
#endif          // an object taller than half a pool: the unpipelined loop
            if (big) {
                int cbase = 0;
                while (todo) {
                    const unsigned long long chunk = next_chunk(todo, cbase, POOL);
                    stage(chunk, cbase, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    composite(chunk, cbase, 0);
                    cbase = __builtin_amdgcn_readlane(cum, 63 - __builtin_clzll(chunk));
                    todo &= ~chunk;
                }
            } else if (todo) {
                int cbase = 0, buf = 0;
                unsigned long long cur = next_chunk(todo, 0, HALF);
                stage(cur, 0, 0);
                while (cur) {
                    todo &= ~cur;
                    const int nbase = __builtin_amdgcn_readlane(cum, 63 - __builtin_clzll(cur));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // `cur` has landed
                    const unsigned long long nxt = todo ? next_chunk(todo, nbase, HALF) : 0ull;
                    if (nxt) stage(nxt, nbase, (buf ^ 1) * HALF);              // in flight while `cur` is composited
                    composite(cur, cbase, buf * HALF);
                    cur = nxt; cbase = nbase; buf ^= 1;
                }
            }
            __syncthreads();
        }
        // ---- 5. the tile's pixels
        float bce = 0.f;
        if (inside) {
            const float D = den + (float)HW * 1e-9f;   // every object adds 1e-9 (models.py:527)
            const float invD = 1.f / D;
            const float pre_ = num * invD;
            const float r = fminf(fmaxf(pre_, 0.f), 1.f);
            recon[pi] = r;
            // torch BCE: log clamped at -100; backward denominator max(r(1-r), 1e-12)
            bce = -(xv * fmaxf(logf(r), -100.f) + (1.f - xv) * fmaxf(logf(1.f - r), -100.f));
            if (aux) {
                const float gr = (pre_ >= 0.f && pre_ <= 1.f) ? (r - xv) / fmaxf(r * (1.f - r), 1e-12f) : 0.f;
                aux[pi] = make_float2(gr * invD, pre_);
            }
        }
        // one partial per tile, at the index k_render_fwd3's one-tile workgroups use (loss_finalize sums them in index order)
        bce = block_reduce_sum_256(bce, red);
        if (tid == 0) {
            const int tile = (ty0 / RT) * tiles_x + (tx0 / RT);
            bce_partial[(size_t)b * tiles_x * tiles_x + tile] = bce;
        }
    }
}

}  // namespace

// SPAIR_ERR_UNSUPPORTED: the caller keeps k_render_fwd3 (render2.hip)
int render_fwd3(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x,
                float* recon, float* aux, float* bce_partial, int B, int HW, int I, int P, int ac, int s_bf16, hipStream_t s) {
    const int texb = s_bf16 ? 4 : 8, es = s_bf16 ? 2 : 4;
    if ((P * texb) % 16 != 0 || ((size_t)ld_s * es) % 16 != 0 || P + 1 > R6_ROWS || P < 2 || P > 255) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * ld_s * es >= (1ull << 32)) return SPAIR_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(S) & 15) != 0 || HW > 1024) return SPAIR_ERR_UNSUPPORTED;
    const int t = (I + RT - 1) / RT, rg = (t + R6_RG - 1) / R6_RG;
    const dim3 grid(B * rg * rg), block(256);
    float2* aux2 = reinterpret_cast<float2*>(aux);
    const size_t lds = (size_t)r6_shared_bytes() + 4 * (size_t)r6_wave_bytes(P, texb);
    if (lds > 160 * 1024) return SPAIR_ERR_UNSUPPORTED;
    const bool ip2 = (I & (I - 1)) == 0;
#define R6_LAUNCH(S16_, PT_, AC_, IP2_, NR_)                                                                                            \
    do {                                                                                                                                \
        static std::atomic<unsigned long long> attr_done{0};                                                                            \
        if (lds > 64 * 1024 &&                                                                                                          \
            spair_dyn_lds_once(reinterpret_cast<const void*>(&k_render_fwd6<S16_, PT_, AC_, IP2_, NR_>), (int)lds, attr_done) != SPAIR_OK) \
            return SPAIR_ERR_LAUNCH;                                                                                                    \
        hipLaunchKernelGGL((k_render_fwd6<S16_, PT_, AC_, IP2_, NR_>), grid, block, lds, s, S, ld_s, nbox, pres, depth, ld_pd, x, recon, \
                           aux2, bce_partial, B, HW, I, P);                                                                             \
    } while (0)
#define R6_PICK(S16_, NR_)                                                                                                              \
    do {                                                                                                                                \
        if (P == 28 && !ac && ip2) R6_LAUNCH(S16_, 28, 0, 1, NR_);                                                                      \
        else if (ac) R6_LAUNCH(S16_, 0, 1, 0, NR_);                                                                                     \
        else R6_LAUNCH(S16_, 0, 0, 0, NR_);                                                                                             \
    } while (0)
    if (s_bf16) { if (HW <= 256) R6_PICK(true, 1); else R6_PICK(true, 4); }
    else { if (HW <= 256) R6_PICK(false, 1); else R6_PICK(false, 4); }
#undef R6_PICK
#undef R6_LAUNCH
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
