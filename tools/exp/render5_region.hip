// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3): "every sprite staged once per 128 x 128 region" forward renderer.  Correct (all renderer / engine
// parity tests passed with it as the default) and bit-deterministic, but SLOWER than k_render_fwd3: 0.254-0.313 ms against 0.207 ms in the step at
// BASELINE configs[1], 0.565 against 0.288 ms at configs[3].  What was measured on the way: 8 computing + 4 loader waves 0.341 ms; 16 computing
// waves sharing the DMA duty + per-wave y tables 0.313; objects visited in a strided (spatially scattered) order instead of cell order 0.260
// (in cell order only the 2-3 waves owning a corner of the image had work in a chunk); chunk-level ballot instead of a serial list walk 0.254.
// The staging is indeed gone (803 LDS-DMA instructions per workgroup), what replaced it is imbalance: a wave owns fixed pixels, a chunk of 8
// objects gives it 1.7 +- 1.2 of them, and the ring (4 slots of 8 sprites in 100 KB of LDS) only lets a wave run two chunks ahead of the slowest.
// To use it: copy to spair_pytorch_amd/csrc/render5.hip, declare render_fwd5 in render2.hip and call it at the top of render_fwd2.
// K6 forward, third structure (reference: models.py:485-547, stn(inverse=True) modules.py:256-269): every sprite is staged into LDS ONCE per
// 128 x 128 image region.
//
// k_render_fwd3 (render2.hip) works per 16 x 16 tile and re-stages an object's sprite rows for every strip of every tile it touches: 0.9 GB
// of L2 -> LDS traffic for 0.2 GB of sprites, and that staging (each CU takes in ~30 B/clk) was 0.12 of its 0.19 ms.  Here one workgroup owns
// a whole region (the whole image at 128 x 128):
//   * a prologue culls the sample's objects against the region and leaves their exact inverse-affine parameters + pixel bounding boxes as an
//     ordered list in LDS;
//   * the listed sprites (3,136 contiguous bytes each, 4 LDS-DMA instructions) stream through a ring of 4 x 8 sprites; each of the 16 waves
//     stages one sprite of every other chunk (803 DMA instructions per workgroup in all: no dedicated loader waves needed);
//   * every wave owns a 16 x 64 pixel half strip of the region with its 16 pixels per lane's (numerator, denominator) accumulators in
//     REGISTERS, walks the ring in list order (so every pixel adds its objects in the order every other kernel uses: results are
//     bit-identical to k_render_fwd3's) and composites the objects whose bounding box meets the half strip: x taps of the lane's column in
//     registers, y taps of the 64 rows in a wave-private table, four 4-byte taps from the sprite in LDS per (pixel, object).
// No workgroup barrier in the loop: a slot's state is two LDS counters (landed: +1 per staging wave whose DMA has completed; freed: +1 per
// wave that is done with it), so a wave with few objects in one chunk runs ahead of one with many.
#include <stdlib.h>
#include "render_common.h"

namespace {

constexpr int R5_RW = 128, R5_RH = 128;          // region
constexpr int R5_WAVES = 16, R5_THREADS = R5_WAVES * 64;
constexpr int R5_CHUNK = 8, R5_NS = 4;           // sprites per ring slot, slots
constexpr int R5_MAXOBJ = 640;                   // list capacity (objects whose footprint meets the region; 48-px boxes on a 256-px image: <= 502)
constexpr int R5_P = 28, R5_SPB = R5_P * R5_P * 4;                 // fp16 (grey, alpha) texels: 3,136 B per sprite
constexpr int R5_HH = 64, R5_PASSES = R5_HH / 4; // a wave owns a 16 x 64 pixel half strip: 16 passes of 4 rows

struct R5Cand {                                  // 48 B
    float ax, bx, ay, by, pres, pd;
    int row, x0, x1, y0, y1, pad;                // pixel bounding box in REGION coordinates (conservative), row of the sprite array
};

constexpr int R5_OFF_CAND = R5_NS * R5_CHUNK * R5_SPB;
constexpr int R5_OFF_YTAB = R5_OFF_CAND + R5_MAXOBJ * 48;          // [16 waves][64 rows] float4
constexpr int R5_OFF_FLAGS = R5_OFF_YTAB + R5_WAVES * R5_HH * 16;
__host__ __device__ constexpr int r5_lds_bytes() { return R5_OFF_FLAGS + 512; }

// One axis of the bilinear footprint (render2.hip rf_axis, duplicated so that the two translation units stay independent)
__device__ __forceinline__ void r5_axis(float s, int P, int& i0, float& w0, float& w1) {
    const bool cov = s > -1.f && s < (float)P;
    const float f0 = floorf(s);
    const float f = s - f0;
    int i = (int)fminf(fmaxf(f0, -1.f), (float)(P - 1));
    float a = 1.f - f, b = f;
    if (i < 0) { a = b; b = 0.f; i = 0; }
    else if (i >= P - 1) { b = a; a = 0.f; i = P - 2; }
    if (!cov) { a = 0.f; b = 0.f; i = 0; }
    i0 = i; w0 = a; w1 = b;
}
template <int AC, int IP2>
__device__ __forceinline__ float r5_base(int j, int n, float inv_n) {
    if constexpr (IP2 && !AC) return (2.f * (float)j + 1.f) * inv_n - 1.f;
    else return stn_base(j, n, AC);
}
__device__ __forceinline__ int r5_lds_load(const volatile int* p) { return *p; }

template <int AC, int IP2>
__global__ __launch_bounds__(R5_THREADS, 4) void k_render_fwd5(const _Float16* __restrict__ S, int ld_s, const float* __restrict__ nbox,
                                                               const float* __restrict__ pres, const float* __restrict__ depth, int ld_pd,
                                                               const float* __restrict__ x, float* __restrict__ recon, float2* __restrict__ aux,
                                                               float* __restrict__ bce_partial, int n_partial, int B, int HW, int I, int kstride) {
    extern __shared__ __attribute__((aligned(16))) char r5_sm[];
    char* ring = r5_sm;                                                        // [NS][CHUNK][3136]
    R5Cand* cand = reinterpret_cast<R5Cand*>(r5_sm + R5_OFF_CAND);
    int* flags = reinterpret_cast<int*>(r5_sm + R5_OFF_FLAGS);                 // [0..3] landed, [4..7] freed, [16..31] per-wave counts, [32..47] bce
    constexpr int P = R5_P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float4* ytab = reinterpret_cast<float4*>(r5_sm + R5_OFF_YTAB) + wave * R5_HH;
    const int regs_x = (I + R5_RW - 1) / R5_RW, regs = regs_x * ((I + R5_RH - 1) / R5_RH);
    const int b = blockIdx.x / regs, reg = blockIdx.x - b * regs;
    const int rx0 = (reg % regs_x) * R5_RW, ry0 = (reg / regs_x) * R5_RH;
    const int rx1 = min(rx0 + R5_RW, I) - 1, ry1 = min(ry0 + R5_RH, I) - 1;
    const float inv_I = 1.f / (float)I;

    // ---- prologue: cull + exact parameters, ordered list in LDS
    if (tid < 16) flags[tid] = 0;
    __syncthreads();
    int nlist = 0;
    // Objects are visited in the fixed order k = (i * kstride) mod HW, i = 0, 1, ... (kstride coprime to HW): the natural order follows the cell
    // grid, so 8 consecutive objects sit in the same corner of the image and only the 2-3 waves that own those pixels would have work per chunk.
    // (Every pixel still adds its objects in ONE fixed order, the same in every run and for every batch split.)
    for (int k0 = 0; k0 < HW; k0 += R5_THREADS) {
        const int ki = k0 + tid;
        bool hit = false;
        R5Cand c;
        if (ki < HW) {
            const int k = (int)(((long long)ki * kstride) % HW);
            const int r = k * B + b;
            const float4 nb = *reinterpret_cast<const float4*>(nbox + (size_t)r * 4);
            const float tx = 2.f * nb.x - 1.f, ty = 2.f * nb.y - 1.f;
            c.ax = 1.f / nb.z; c.bx = -tx / nb.z; c.ay = 1.f / nb.w; c.by = -ty / nb.w;          // the same expressions as the backward's
            float cx0, sxl, cy0, syl;
            src_affine(c.ax, c.bx, I, P, AC, cx0, sxl);
            src_affine(c.ay, c.by, I, P, AC, cy0, syl);
            // pixels whose source coordinate lies in (-1, P), one pixel of slack on both sides (a superset only costs zero weights)
            const float ix = __builtin_amdgcn_rcpf(sxl), iy = __builtin_amdgcn_rcpf(syl);
            const int X0 = (int)floorf((-1.f - cx0) * ix) - 1, X1 = (int)ceilf(((float)P - cx0) * ix) + 1;
            const int Y0 = (int)floorf((-1.f - cy0) * iy) - 1, Y1 = (int)ceilf(((float)P - cy0) * iy) + 1;
            hit = X1 >= rx0 && X0 <= rx1 && Y1 >= ry0 && Y0 <= ry1;
            c.pres = pres[(size_t)r * ld_pd];
            c.pd = c.pres * depth[(size_t)r * ld_pd];
            c.row = r; c.pad = 0;
            c.x0 = max(X0, rx0) - rx0; c.x1 = min(X1, rx1) - rx0; c.y0 = max(Y0, ry0) - ry0; c.y1 = min(Y1, ry1) - ry0;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) flags[16 + wave] = __popcll(bal);
        __syncthreads();
        int base = nlist;
        for (int w = 0; w < wave; ++w) base += flags[16 + w];
        int tot = 0;
        for (int w = 0; w < R5_WAVES; ++w) tot += flags[16 + w];
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && slot < R5_MAXOBJ) cand[slot] = c;
        nlist = min(nlist + tot, R5_MAXOBJ);
        __syncthreads();
    }
    const int nch = (nlist + R5_CHUNK - 1) / R5_CHUNK;

    // ---- DMA duty: wave w stages sprite (w & 7) of the chunks with parity (w >> 3): 4 LDS-DMA instructions per sprite (3,136 contiguous bytes).
    // A slot's state is two LDS counters -- landed (+1 per staging wave whose DMA has completed) and freed (+1 per wave that is done with it):
    // no workgroup barrier in the loop, a wave with few objects in a chunk runs up to two chunks ahead of one with many.
    const char* Sb = reinterpret_cast<const char*>(S);
    const int my_par = wave >> 3, my_oi = wave & 7;
    auto issue = [&](int c) {
        const int slot = c % R5_NS, gen = c / R5_NS;
        if (gen > 0) {                           // every wave has released the slot's previous chunk
            while (r5_lds_load(flags + 4 + slot) < R5_WAVES * gen) __builtin_amdgcn_s_sleep(1);
        }
        const int idx = min(c * R5_CHUNK + my_oi, nlist - 1);              // past the end: the last sprite again (never read)
        const unsigned so = (unsigned)cand[idx].row * (unsigned)(ld_s * 2);
        char* dst = ring + (slot * R5_CHUNK + my_oi) * R5_SPB;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p * 1024 + lane * 16 < R5_SPB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Sb + (size_t)so + p * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
        }
    };
    if (my_par < nch && my_par == 0) issue(0);
    if (1 < nch && my_par == 1) issue(1);

    const int sx0 = (wave & 7) * 16, hy0 = (wave >> 3) * R5_HH;   // this wave's half strip: region columns sx0 .. +15, rows hy0 .. +63
    const int col = lane & 15, rg = lane >> 4;
    const int px = rx0 + sx0 + col;
    const float basex = r5_base<AC, IP2>(min(px, I - 1), I, inv_I);
    float num[R5_PASSES], den[R5_PASSES];
#pragma unroll
    for (int p = 0; p < R5_PASSES; ++p) { num[p] = 0.f; den[p] = 0.f; }
    for (int c = 0; c < nch; ++c) {
        const int slot = c % R5_NS, gen = c / R5_NS;
        const bool mine = (c & 1) == my_par;
        if (mine) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // chunk c's four instructions (issued two iterations ago) have landed
            if (lane == 0) __hip_atomic_fetch_add(flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (c + 2 < nch) issue(c + 2);                                     // (after the publication: issue() may have to wait for a slow wave)
        }
        while (r5_lds_load(flags + slot) < R5_CHUNK * (gen + 1)) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const int n_here = min(R5_CHUNK, nlist - c * R5_CHUNK);
        // which of the chunk's objects meet this half strip: one object per lane, one ballot (a serial walk over the list paid an LDS round
        // trip per object and wave, met or not)
        unsigned long long todo;
        {
            bool meet = false;
            if (lane < n_here) {
                const R5Cand& cl = cand[c * R5_CHUNK + lane];
                meet = cl.x1 >= sx0 && cl.x0 <= sx0 + 15 && cl.y1 >= hy0 && cl.y0 <= hy0 + R5_HH - 1;
            }
            todo = __ballot(meet);
        }
        for (; todo; todo &= todo - 1) {
            const int o = __builtin_ctzll(todo);
            const R5Cand& cd = cand[c * R5_CHUNK + o];
            const int oy0 = __builtin_amdgcn_readfirstlane(cd.y0) - hy0, oy1 = __builtin_amdgcn_readfirstlane(cd.y1) - hy0;
            const float ax = cd.ax, bx = cd.bx, ay = cd.ay, by = cd.by, prs = cd.pres, pd = cd.pd;
            float gd;
            // y taps of the half strip's 64 rows, one row per lane, into the wave's table; x taps of the lane's column in registers
            {
                const int py = ry0 + hy0 + lane;
                int j0;
                float wy0, wy1;
                r5_axis(src_from_base(ay, by, r5_base<AC, IP2>(min(py, I - 1), I, inv_I), P, AC, gd), P, j0, wy0, wy1);
                ytab[lane] = make_float4(__int_as_float(j0 * (P * 4)), wy0, wy1, 0.f);
            }
            int i0x;
            float wx0, wx1;
            r5_axis(src_from_base(ax, bx, basex, P, AC, gd), P, i0x, wx0, wx1);
            const char* sp = ring + (slot * R5_CHUNK + o) * R5_SPB + i0x * 4;
            wave_lds_fence();
#pragma unroll
            for (int p = 0; p < R5_PASSES; ++p) {
                if (4 * p + 3 < oy0 || 4 * p > oy1) continue;             // wave-uniform
                const float4 ye = ytab[4 * p + rg];
                const char* tp = sp + __float_as_int(ye.x);
                const float wy0 = ye.y, wy1 = ye.z;
                const float2 t00 = sprite_unpack(*reinterpret_cast<const unsigned*>(tp)), t01 = sprite_unpack(*reinterpret_cast<const unsigned*>(tp + 4));
                const float2 t10 = sprite_unpack(*reinterpret_cast<const unsigned*>(tp + P * 4)), t11 = sprite_unpack(*reinterpret_cast<const unsigned*>(tp + P * 4 + 4));
                // (the expression order of k_render_fwd3's composite loop, term by term)
                const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
                float g = w00 * t00.x, al = w00 * t00.y, mm = w00 * fmaxf(t00.y * pd, 0.01f);
                g = fmaf(w01, t01.x, g); al = fmaf(w01, t01.y, al); mm = fmaf(w01, fmaxf(t01.y * pd, 0.01f), mm);
                g = fmaf(w10, t10.x, g); al = fmaf(w10, t10.y, al); mm = fmaf(w10, fmaxf(t10.y * pd, 0.01f), mm);
                g = fmaf(w11, t11.x, g); al = fmaf(w11, t11.y, al); mm = fmaf(w11, fmaxf(t11.y * pd, 0.01f), mm);
                al *= prs;
                num[p] += g * al * (mm + 1e-9f);
                den[p] += mm;
            }
            wave_lds_fence();                                                  // the table is rewritten for the next object
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // this wave's reads of the slot are done
        if (lane == 0) __hip_atomic_fetch_add(flags + 4 + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- epilogue: normalise, clamp, BCE (models.py:527-547) for the lane's 16 pixels
    float bce = 0.f;
#pragma unroll
    for (int p = 0; p < R5_PASSES; ++p) {
        const int py = ry0 + hy0 + 4 * p + rg;
        if (px < I && py < I) {
            const size_t pi = ((size_t)b * I + py) * I + px;
            const float xv = x[pi];
            const float D = den[p] + (float)HW * 1e-9f;                      // every object adds 1e-9 (models.py:527)
            const float invD = 1.f / D;
            const float pre = num[p] * invD;
            const float r = fminf(fmaxf(pre, 0.f), 1.f);
            recon[pi] = r;
            bce += -(xv * fmaxf(logf(r), -100.f) + (1.f - xv) * fmaxf(logf(1.f - r), -100.f));
            if (aux) {
                const float gr = (pre >= 0.f && pre <= 1.f) ? (r - xv) / fmaxf(r * (1.f - r), 1e-12f) : 0.f;
                aux[pi] = make_float2(gr * invD, pre);
            }
        }
    }
    bce = wave_reduce_sum(bce);
    if (lane == 0) reinterpret_cast<float*>(flags)[32 + wave] = bce;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < R5_WAVES; ++w) t += reinterpret_cast<float*>(flags)[32 + w];     // fixed order
        bce_partial[blockIdx.x] = t;
        for (int i = blockIdx.x + gridDim.x; i < n_partial; i += gridDim.x) bce_partial[i] = 0.f;      // the caller sums n_partial entries
    }
}

}  // namespace

// 16-bit (fp16 pair) sprites, P = 28 only; SPAIR_ERR_UNSUPPORTED: the caller falls back to k_render_fwd3.  n_partial: entries of bce_partial
// the caller will sum (>= the number of workgroups; the surplus is zeroed here).
int render_fwd5(const float* S, int ld_s, const float* nbox, const float* pres, const float* depth, int ld_pd, const float* x, float* recon,
                float* aux, float* bce_partial, int n_partial, int B, int HW, int I, int P, int ac, int s_bf16, hipStream_t s) {
    if (!s_bf16 || P != R5_P || (ld_s & 7) || (reinterpret_cast<uintptr_t>(S) & 15)) return SPAIR_ERR_UNSUPPORTED;
    if ((unsigned long long)B * HW * ld_s * 2 >= (1ull << 32)) return SPAIR_ERR_UNSUPPORTED;
    const int regs = ((I + R5_RW - 1) / R5_RW) * ((I + R5_RH - 1) / R5_RH);
    const int grid = B * regs;
    if (grid > n_partial) return SPAIR_ERR_UNSUPPORTED;
    // objects whose footprint can meet one region: all HW of them in the worst case
    if (HW > R5_MAXOBJ && regs == 1) return SPAIR_ERR_UNSUPPORTED;
    int kstride = 1;                             // smallest odd number >= 0.38 HW that is coprime to HW
    {
        auto gcd = [](int a, int b_) { while (b_) { const int t = a % b_; a = b_; b_ = t; } return a; };
        for (int sdd = (int)(0.38 * HW) | 1; sdd < HW; sdd += 2)
            if (gcd(sdd, HW) == 1) { kstride = sdd; break; }
    }
    const int lds = r5_lds_bytes();
    const bool ip2 = (I & (I - 1)) == 0;
    const _Float16* Sh = reinterpret_cast<const _Float16*>(S);
    float2* aux2 = reinterpret_cast<float2*>(aux);
#define R5_LAUNCH(AC_, IP2_)                                                                                                              \
    do {                                                                                                                                  \
        static bool attr_set = false;                                                                                                     \
        if (!attr_set) {                                                                                                                  \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_fwd5<AC_, IP2_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    lds) != hipSuccess)                                                                                   \
                return SPAIR_ERR_LAUNCH;                                                                                                  \
            attr_set = true;                                                                                                              \
        }                                                                                                                                 \
        hipLaunchKernelGGL((k_render_fwd5<AC_, IP2_>), dim3(grid), dim3(R5_THREADS), lds, s, Sh, ld_s, nbox, pres, depth, ld_pd, x, recon,   \
                           aux2, bce_partial, n_partial, B, HW, I, kstride);                                                                       \
    } while (0)
    if (ac) R5_LAUNCH(1, 0);
    else if (ip2) R5_LAUNCH(0, 1);
    else R5_LAUNCH(0, 0);
#undef R5_LAUNCH
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
