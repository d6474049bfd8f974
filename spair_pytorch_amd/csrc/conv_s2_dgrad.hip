#include <numeric>
// Patch-resident DATA GRADIENT of the backbone's strided convolutions in the bf16 step (the backward of modules.py:59-64's Conv2d(128, 128, 4,
// stride 2) + the ReLU gate of the layer below), optionally with the stem's weight gradient taken from the gated tile (conv_1: d act0 is never
// written to HBM, as in gemm16.hip's STEM path).
//
// The data gradient of a 4x4 / stride-2 convolution splits into 4 output-parity classes (py, px): the pixels (2y + py, 2x + px) of the layer's
// input receive  sum over the 2x2 taps (ty, tx) and co of  d_out[y - ty][x - tx][co] . W[co][ci][py + 2 ty][px + 2 tx]  -- a 2x2, stride-1
// convolution of d_out per class.  All 4 classes x 4 taps read the SAME d_out neighbourhood, so a workgroup stages the zero-bordered d_out
// patch of its 128 class pixels ONCE (both 64-channel halves, <= 7 rows x 36 pixels x 256 B at conv_1) and runs the 16 (class, tap) products
// -- 32 K steps of 64 -- out of it: the gathered operand costs 64 KB per tile instead of 4 x 128 KB through the implicit-GEMM kernel, which was
// bound by exactly those bytes (conv_1's data gradient: 0.31 ms for 155 GFLOP).  Only the 16 KB weight tile of each K step streams (ring of 3).
// Wave roles as in conv_s2.hip: 8 computing waves (a 64-channel x 32-pixel block of the transposed tile each, two per SIMD; round 6 -- 4 waves of
// 64 x 64, one per SIMD, before) + 4 loader waves that issue every LDS-DMA behind counted waits.
// Per class an epilogue on the computing waves: accumulators -> bf16 tile in LDS -> whole 256-byte rows: ReLU gate (the stored activation of the
// layer below > 0), then either the row-mapped store of d act or (STEM) the gated tile x the 4x4 input patches on the matrix cores, summed
// over the tile's 4 classes and left as ONE [128][17] partial per workgroup.
#include "common.h"
#include "gemm.h"

namespace {

constexpr int DG_BM = 128, DG_C = 128;
constexpr int DG_PPX = 256;                      // patch capacity in pixels per 64-channel half (32 DMA pieces of 8 pixels x 128 B)
constexpr int DG_PATCH_B = DG_PPX * 128;         // 32 KB
constexpr int DG_BT_B = 128 * 128;               // weight tile [128 ci][64 k] bf16
constexpr int DG_LDG = DG_C + 8, DG_LDP = 32 + 8;
constexpr int DG_OFF_PATCH = 3 * DG_BT_B;
constexpr int DG_OFF_G = DG_OFF_PATCH + 2 * DG_PATCH_B;            // gated / staged tile [128][136] bf16
constexpr int DG_OFF_P = DG_OFF_G + DG_BM * DG_LDG * 2;            // stem patches [128][40] bf16
constexpr int DG_OFF_DUMP = DG_OFF_P + DG_BM * DG_LDP * 2;
constexpr int DG_LDS = DG_OFF_DUMP + 1024;                         // 160,768 B
constexpr int DG_STEM_FLOATS = 128 * 17;

struct ConvDgradArgs {
    const u16* dout; const u16* Bz[4]; const u16* gate; u16* out;
    const unsigned char* gbits;                  // BITS: the gate as sign bits, one byte per (pixel, 8 channels) [B][Hi][Hi][16] (the stem kernel's mask)
    int B, Ho, Hc, Hi, M;                        // d_out side, class-grid side Ho + 1, input side 2 Hc, M = B * Hc * Hc
    int tpi;                                     // 0: tiles of 128 consecutive class pixels of the whole batch; > 0: tiles per image
    // STEM (conv_1): xp = zero-padded fp32 stem input [B][stem_hin][stem_hin], stride stem_s; one [128][17] partial per workgroup
    const float* stem_xp; float* stem_part; int stem_hin, stem_s;
};

// Swizzle key of weight-tile row n (16-byte chunk c of the row sits at position c ^ key): the computing waves read the rows in the PERMUTED
// order n(T, m) = 32 (T >> 1) + 8 (m >> 2) + 4 (T & 1) + (m & 3) (so that a lane ends up with 8 consecutive output channels, see the kernel);
// with the usual key n & 7 the 16 rows of a fragment would only produce 4 distinct keys.  This one gives 8, conflict-free for ds_read_b128.
__device__ __forceinline__ int dg_wkey(int n) { return (n & 3) | (((n >> 3) & 1) << 2); }
template <int W>
__device__ __forceinline__ void dg_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory"); }

typedef short dg_v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 dg_tr_frag16(const __bf16* tile, int ld, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a0 = tile + (8 * g + q) * ld + c0 + 4 * p;
    const __bf16* a1 = a0 + 4 * ld;
    const dg_v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dg_v4s16*)(a0));
    const dg_v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) dg_v4s16*)(a1));
    union { struct { dg_v4s16 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

template <bool STEM, bool BITS>
__global__ __launch_bounds__(768, 3) void k_conv_s2k4_dgrad(ConvDgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dg_sm[];
    char* bt = dg_sm;                            // [3][128 ci][64 k] bf16, chunk ^ (row & 7)
    char* patch = dg_sm + DG_OFF_PATCH;          // [2 halves][DG_PPX][64 ch] bf16, chunk ^ (pixel & 7)
    __bf16* Gs = reinterpret_cast<__bf16*>(dg_sm + DG_OFF_G);
    __bf16* Ps = reinterpret_cast<__bf16*>(dg_sm + DG_OFF_P);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int Ho = a.Ho, Hc = a.Hc, Wp = Ho + 2, Hp = Ho + 2;
    // PERSISTENT: one workgroup per CU walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (a tile is short-lived -- 32 K steps -- against
    // the 3-4 us a 160-KB workgroup takes to launch and fill its first patch; the loaders start the next tile's patch and weight tiles
    // behind barrier(32), under the computing waves' last class epilogue)
    const int ntiles = a.tpi ? a.B * a.tpi : (a.M + DG_BM - 1) / DG_BM;
    int m0 = 0, mend = 0, L0 = 0, npx = 0;
    // class pixel m = (b, y, x) on the Hc x Hc class grid; g = b * Hc + y.  d_out rows with a zero border: extended row E = b * Hp + yd + 1;
    // tap (ty, tx) of class pixel (g, x) reads extended pixel (g + b + 1 - ty) * Wp + x + 1 - tx, Wp = Ho + 2.  The patch is the LINEAR window
    // of extended pixels from the first tile pixel's tap (1, 1) to the last tile pixel's tap (0, 0): 128 + Wp + 1 pixels plus a row per
    // image boundary crossed (whole rows E0 .. E1 refused conv_1 at 256 x 256: 4-5 rows of 68).
    auto set_tile = [&](int tile) {
        if (a.tpi) {
            const int img = tile / a.tpi, HH = Hc * Hc;
            m0 = img * HH + (tile - img * a.tpi) * DG_BM;
            mend = min(m0 + DG_BM, (img + 1) * HH);
        } else {
            m0 = tile * DG_BM;
            mend = min(m0 + DG_BM, a.M);
        }
        const int mlast = mend - 1;
        const int g0 = m0 / Hc, g1 = mlast / Hc;
        L0 = (g0 + g0 / Hc) * Wp + (m0 - g0 * Hc);
        npx = (g1 + g1 / Hc + 1) * Wp + (mlast - g1 * Hc) + 1 - L0 + 1;      // <= DG_PPX (checked by the launcher)
    };
    // 8 computing waves (two per SIMD: one's LDS round trips and class epilogue meet the other's MFMAs -- with one per SIMD every
    // fragment-read latency and the whole register epilogue of a class were exposed: MFMA pipe 0.33 busy) + 4 loader waves (one per SIMD)
    const bool loader = wave >= 8;
    const int lw = wave & 3;

    if (loader) {
        const __amdgpu_buffer_rsrc_t rin = buf_rsrc(a.dout);
        auto glds = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte_off, char* dst) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)byte_off, 0, 0, 0);
        };
        auto issue_b = [&](int kt) {             // K step kt = (class, half, tap): weight tile -> ring slot kt % 3; 4 pieces (8 rows each) per loader
            const int cls = (kt >> 3) & 3, half = (kt >> 2) & 1, tap = kt & 3;
            const bool live = kt < 32;
            const __amdgpu_buffer_rsrc_t rw = buf_rsrc(a.Bz[cls]);
            char* dst = bt + (kt % 3) * DG_BT_B;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int piece = lw * 4 + p, n = piece * 8 + (lane >> 3), pos = lane & 7, c = pos ^ dg_wkey(n);
                glds(rw, live ? ((unsigned)n * 512u + (unsigned)(tap * 128 + half * 64 + c * 8)) * 2u : BUF_OOB, dst + piece * 1024);
            }
        };
        auto issue_patch = [&](int half, int part, int npieces) {      // pieces (part * 4 + lw) * npieces .. of the d_out patch's channel half
            char* dst = patch + half * DG_PATCH_B;
            for (int e = 0; e < npieces; ++e) {
                const int piece = (part * 4 + lw) * npieces + e;
                const int pp = piece * 8 + (lane >> 3), pos = lane & 7, c = pos ^ (pp & 7);
                const int pg = L0 + pp, E = pg / Wp, xp = pg - E * Wp;
                const int b = E / Hp, yd = E - b * Hp - 1, xd = xp - 1;
                const bool ok = pp < npx && b < a.B && yd >= 0 && yd < Ho && xd >= 0 && xd < Ho;
                const unsigned off = (((unsigned)(b * Ho + yd) * (unsigned)Ho + (unsigned)xd) * DG_C + half * 64 + c * 8) * 2u;
                glds(rin, ok ? off : BUF_OOB, piece < DG_PPX / 8 ? dst + piece * 1024 : dg_sm + DG_OFF_DUMP);
            }
        };
#pragma unroll 1
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        set_tile(tile);
        issue_patch(0, 0, 8);                    // half 0: 32 pieces = 4 loaders x 8
        issue_b(0);
        issue_b(1);
        // STEM: barrier E of the PREVIOUS tile's last class is joined here, behind the issue and in front of the wait -- the computing waves run
        // that epilogue meanwhile and would otherwise sit at E until this tile's patch has landed
        if (STEM && tile != (int)blockIdx.x) __builtin_amdgcn_s_barrier();
        dg_wait<4>();                            // everything but tile 1
#pragma unroll 1
        for (int kt = 0; kt < 32; ++kt) {
            __builtin_amdgcn_s_barrier();        // barrier(kt): the computing waves have read all of tile kt - 1 (ring slot (kt + 2) % 3)
            asm volatile("" ::: "memory");
            if (kt < 2) {                        // half 1 (first read at step 4): 2 x 4 pieces per loader, in front of the weight tile
                issue_patch(1, kt, 4);
                issue_b(kt + 2);
                dg_wait<8>();                    // tile kt + 1 = the last 4 operations of the previous batch; younger: this batch (4 + 4)
            } else {
                issue_b(kt + 2);
                dg_wait<4>();
            }
            if (STEM && kt != 0 && (kt & 7) == 0) __builtin_amdgcn_s_barrier();      // barrier E of the class that ended with step kt - 1 (the
                                                                                     // computing waves run its epilogue behind barrier(kt))
        }
        __builtin_amdgcn_s_barrier();            // barrier(32): every read of this tile's ring and patch is in registers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy tail DMAs target ring slots the next tile fills (and this workgroup's LDS)
        }
        if (STEM) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }      // E of the last tile's last class, X, Y
        return;
    }

    // ---------------------------------------------------------------- computing waves (tid 0 .. 511)
    // The products are computed TRANSPOSED, C'[ci][pixel] = sum_k W[ci][k] . patch[pixel][k] (weights = A operand, patch = B operand): a lane of the
    // accumulator then holds 4 consecutive channels of ONE pixel, and with the weight rows of a tile pair read in the order n(T, m) above 8
    // consecutive ones -- exactly a 16-byte piece of the pixel's NHWC row.  The ReLU gate and the store (or the write of the gated tile for the
    // stem product) work on registers: no staging of the accumulators through LDS, no epilogue barrier in the plain kernel (staged through a
    // bf16 tile with 2-byte LDS writes the epilogue of a class took 4,500 cycles, as long as 4 of its 8 K steps, and stalled the loaders).
    // wave -> (channel half wn, pixel half wm, 32-pixel quarter wp): a 64-channel x 32-pixel block of the transposed tile per wave
    const int wn = wave & 1, wm = (wave >> 1) & 1, wp = wave >> 2;
    const int prow0 = wm * 64 + wp * 32;         // the wave's first tile pixel
    int pbase[2];                                // patch pixel of (pixel tile j, pixel r16) at tap (0, 0)
    unsigned orow[2];                            // its output pixel of class (0, 0)
    bool ook[2];
    auto set_rows = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + prow0 + j * 16 + r16;
            ook[j] = m < mend;
            const int mc = min(m, mend - 1);
            const int g = mc / Hc, x = mc - g * Hc, b = g / Hc, y = g - b * Hc;
            pbase[j] = (g + b + 1) * Wp + x + 1 - L0;
            orow[j] = (unsigned)((b * a.Hi + 2 * y) * a.Hi + 2 * x);
        }
    };
    // weight-tile rows of this lane: tile t of the wave's 64 channels, lane row m = r16
    const int wkey = (r16 & 3) | (((r16 >> 2) & 1) << 2);
    const int wrow0 = wn * 64 + (r16 >> 2) * 8 + (r16 & 3);          // + 32 (t >> 1) + 4 (t & 1)
    const int ch0 = wn * 64 + q * 8;                                  // + 32 p: the 8 channels this lane owns after tile pair p
    const __amdgpu_buffer_rsrc_t rgate = buf_rsrc(BITS ? reinterpret_cast<const void*>(a.gbits) : reinterpret_cast<const void*>(a.gate)), rout = buf_rsrc(a.out);
    f32x4 sacc[2];                               // STEM: this wave's 16 channels x (16 taps | bias | 0 ...), over the tile's 4 classes
#pragma unroll
    for (int j = 0; j < 2; ++j) sacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 acc[4][2];                             // [channel tile t][pixel tile j]
    bf16x8 wfA[4], pfA[2], wfB[4], pfB[2];
    auto read_frags = [&](int kt, int half, int tap, int ks, bf16x8 (&wf)[4], bf16x8 (&pf)[2]) {
        const char* bs = bt + (kt % 3) * DG_BT_B;
        const char* pb = patch + half * DG_PATCH_B;
        const int toff = -((tap >> 1) * Wp + (tap & 1));
        const int c = ks * 4 + q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = wrow0 + (t >> 1) * 32 + (t & 1) * 4;
            wf[t] = *reinterpret_cast<const bf16x8*>(bs + n * 128 + ((c ^ wkey) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = pbase[j] + toff;
            pf[j] = *reinterpret_cast<const bf16x8*>(pb + pp * 128 + ((c ^ (pp & 7)) << 4));
        }
    };
    auto mma = [&](const bf16x8 (&wf)[4], const bf16x8 (&pf)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], pf[j], acc[t][j], 0, 0, 0);
    };
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    set_tile(tile);
    set_rows();
    __builtin_amdgcn_s_barrier();                // barrier(0): tile 0 and the first patch half have landed
    asm volatile("" ::: "memory");
    read_frags(0, 0, 0, 0, wfA, pfA);
    for (int cls = 0; cls < 4; ++cls) {
        const int py = cls >> 1, px = cls & 1;                         // [channel tile t][pixel tile j]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // gate pieces of this class (and the stem patches) are requested at the top of the class: their latency hides behind its 8 K steps
        uint4 gate[BITS ? 1 : 2][2];
        u32x2_t gbits[2];                        // BITS: the 64 sign bits of this wave half's 64 channels of the pixel; the lane's two bytes are q and q + 4
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned pix = orow[j] + (unsigned)(py * a.Hi + px);
            if (BITS) gbits[j] = __builtin_amdgcn_raw_buffer_load_b64(rgate, ook[j] ? (int)(pix * 16u + (unsigned)(wn * 8)) : (int)BUF_OOB, 0, 0);
            else {
#pragma unroll
                for (int p = 0; p < 2; ++p) gate[j][p] = buf_load16(rgate, ook[j] ? (pix * DG_C + ch0 + p * 32) * 2u : BUF_OOB);
            }
        }
        float2 sp[2];
        bool sp_ok = false;
        if (STEM) {                              // thread (tile pixel prow, kernel row kr): the four stem taps of that row
            const int prow = tid >> 2, kr = tid & 3;
            const int m = m0 + prow;
            sp_ok = m < mend;
            const int mc = min(m, mend - 1);
            const int g = mc / Hc, x = mc - g * Hc, b = g / Hc, y = g - b * Hc;
            const int yy = 2 * y + py, xx = 2 * x + px;
            const float* src = a.stem_xp + ((size_t)b * a.stem_hin + yy * a.stem_s + kr) * a.stem_hin + xx * a.stem_s;
            sp[0] = *reinterpret_cast<const float2*>(src);
            sp[1] = *reinterpret_cast<const float2*>(src + 2);
        }
        // K steps of the class, software-pipelined across the workgroup barriers: the fragments of (kt, ks 1) are read while the MFMAs of
        // (kt, ks 0) run, those of (kt + 1, ks 0) -- right behind barrier(kt + 1), which certifies that tile -- while the MFMAs of (kt, ks 1)
        // run.  Issued just in front of their MFMAs, every one of the 16 fragment reads of a K step exposed its LDS latency (4 waves reading
        // 64 KB per step keep the LDS pipe busy for longer than the 512 MFMA cycles): a K step took 1,675 cycles.
#pragma unroll 1
        for (int h4 = 0; h4 < 8; ++h4) {
            const int kt = cls * 8 + h4;
            read_frags(kt, h4 >> 2, h4 & 3, 1, wfB, pfB);
            __builtin_amdgcn_sched_barrier(0);   // (pins the order: left alone hipcc sinks every read to just in front of its first MFMA)
            mma(wfA, pfA);
            __builtin_amdgcn_sched_barrier(0);
            // every read of tile kt is in registers: its ring slot may be refilled.  (The builtin, not inline asm: hipcc's own wait
            // insertion does not see through asm and would wait for these reads AGAIN in front of the MFMAs below -- with the next
            // tile's reads already queued behind them, i.e. for those too.)
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();        // barrier(kt + 1): tile kt + 1 has landed (barrier(32): nothing follows)
            asm volatile("" ::: "memory");
            read_frags(kt + 1, ((h4 + 1) & 7) >> 2, (h4 + 1) & 3, 0, wfA, pfA);      // (behind barrier(32): stale bytes of slot 2, never used)
            __builtin_amdgcn_sched_barrier(0);
            mma(wfB, pfB);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- class epilogue on registers: gate = stored activation of the layer below > 0 (bf16 sign / zero test), 8 channels of one pixel per lane
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (__bf16)acc[2 * p][j][e]; v[4 + e] = (__bf16)acc[2 * p + 1][j][e]; }
                uint4 vv;
                __builtin_memcpy(&vv, &v, 16);
                const unsigned vw[4] = {vv.x, vv.y, vv.z, vv.w};
                unsigned ow[4];
                if (BITS) {
                    const int gb = (int)((p ? gbits[j].y : gbits[j].x) >> (8 * q));      // (bits 0..7 of it are the lane's 8 channels)
#pragma unroll
                    for (int e = 0; e < 4; ++e)      // v_bfe_i32: bit -> 0 / -1
                        ow[e] = vw[e] & (((unsigned)__builtin_amdgcn_sbfe(gb, 2 * e, 1) & 0xffffu) | ((unsigned)__builtin_amdgcn_sbfe(gb, 2 * e + 1, 1) & 0xffff0000u));
                } else {
                    const unsigned gw[4] = {gate[BITS ? 0 : j][p].x, gate[BITS ? 0 : j][p].y, gate[BITS ? 0 : j][p].z, gate[BITS ? 0 : j][p].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned glo = gw[e] & 0xffffu, ghi = gw[e] >> 16;
                        const unsigned mlo = ((glo & 0x8000u) == 0 && (glo & 0x7fffu) != 0) ? 0xffffu : 0u;
                        const unsigned mhi = ((ghi & 0x8000u) == 0 && (ghi & 0x7fffu) != 0) ? 0xffff0000u : 0u;
                        ow[e] = vw[e] & (mlo | mhi);
                    }
                }
                const uint4 o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (STEM) *reinterpret_cast<uint4*>(Gs + (prow0 + j * 16 + r16) * DG_LDG + ch0 + p * 32) = ook[j] ? o : make_uint4(0u, 0u, 0u, 0u);
                else buf_store16(rout, ook[j] ? ((orow[j] + (unsigned)(py * a.Hi + px)) * DG_C + ch0 + p * 32) * 2u : BUF_OOB, o);
            }
        if (STEM) {
            const int prow = tid >> 2, kr = tid & 3;
            bf16x4 o, one;
            const float pv[4] = {sp[0].x, sp[0].y, sp[1].x, sp[1].y};
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (__bf16)(sp_ok ? pv[e] : 0.f); one[e] = (__bf16)0.f; }
            one[0] = (__bf16)((sp_ok && kr == 0) ? 1.f : 0.f);
            *reinterpret_cast<bf16x4*>(&Ps[prow * DG_LDP + kr * 4]) = o;
            *reinterpret_cast<bf16x4*>(&Ps[prow * DG_LDP + 16 + kr * 4]) = one;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // E: the gated tile and the patches are written
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < DG_BM / 32; ++ks) {
                bf16x8 bfr[2];
                const bf16x8 af = dg_tr_frag16(Gs + ks * 32 * DG_LDG, DG_LDG, wave * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j] = dg_tr_frag16(Ps + ks * 32 * DG_LDP, DG_LDP, j * 16, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], sacc[j], 0, 0, 0);
            }
            // (the next class overwrites the tile / patches only behind its 8 K-step barriers)
        }
    }
    }      // tiles
    if (STEM) {
        __builtin_amdgcn_s_barrier();            // X: every wave is done with the last class's stem product (reads of Gs)
        float* Ws = reinterpret_cast<float*>(Gs);            // [128 ch][17]
        const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = wave * 16 + rgrp + r;
            Ws[ch * 17 + col_l] = sacc[0][r];
            if (col_l == 0) Ws[ch * 17 + 16] = sacc[1][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float4* pt = reinterpret_cast<float4*>(a.stem_part + (size_t)blockIdx.x * DG_STEM_FLOATS);
        for (int qq = tid; qq < DG_STEM_FLOATS / 4; qq += 512) pt[qq] = reinterpret_cast<const float4*>(Ws)[qq];
    }
}

}  // namespace

int spair_stem_fused_reduce(float* part, int nblk, float* dw, float* db, hipStream_t s);      // gemm16.hip

// dout: bf16 NHWC [B][Ho][Ho][128]; wd[q]: bf16 [128 ci][4 taps * 128 co] of output-parity class q = py * 2 + px (k_prep mode 3); gate: the
// stored activation of the layer below, bf16 NHWC [B][2 (Ho + 1)][2 (Ho + 1)][128]; out: d of that activation (same layout), or -- when
// stem_part != nullptr -- nothing: the stem's weight / bias gradient is accumulated into stem_dw [128][16] / stem_db [128] instead.
// SPAIR_ERR_UNSUPPORTED: the caller keeps the implicit-GEMM kernel.
int conv_s2k4_patch_dgrad16(const void* dout, const void* const* wd, const void* gate, void* out, int B, int Ho, int hin, int cin, int cout, int k,
                            int s_, const float* stem_xp, int stem_hin, int stem_s, float* stem_part, long long stem_part_cap, float* stem_dw,
                            float* stem_db, hipStream_t s, const void* gate_bits) {
    const int Hc = Ho + 1;
    if (cin != DG_C || cout != DG_C || k != 4 || s_ != 2 || hin != 2 * Hc || B <= 0 || Ho <= 0) return SPAIR_ERR_UNSUPPORTED;
    const long long M = (long long)B * Hc * Hc;
    if ((long long)B * hin * hin * DG_C >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    auto window = [&](long long m0, long long ml) {
        const long long g0 = m0 / Hc, g1 = ml / Hc;
        return (int)((g1 + g1 / Hc + 1) * (Ho + 2) + (ml - g1 * Hc) + 1 - ((g0 + g0 / Hc) * (Ho + 2) + (m0 - g0 * Hc)) + 1);
    };
    int tiles = (int)((M + DG_BM - 1) / DG_BM), tpi = 0, worst = 0;
    // the tile pattern repeats every HH / gcd(DG_BM, HH) tiles: every distinct tile is checked; a period too long to walk takes the per-image tiling
    const long long HHc = (long long)Hc * Hc, period = HHc / std::gcd((long long)DG_BM, HHc);
    const bool walk = std::min<long long>(tiles, period) <= 65536;
    for (int t = 0; walk && t < tiles && t < period; ++t)
        worst = std::max(worst, window((long long)t * DG_BM, std::min<long long>((long long)(t + 1) * DG_BM, M) - 1));
    if (walk && tiles > period) worst = std::max(worst, window((long long)(tiles - 1) * DG_BM, M - 1));
    if (!walk || worst > DG_PPX) {      // tiles that restart at every image (its last tile partial)
        const int HH = Hc * Hc;
        tpi = (HH + DG_BM - 1) / DG_BM;
        worst = 0;
        for (int t = 0; t < tpi; ++t) worst = std::max(worst, window((long long)t * DG_BM, std::min<long long>((long long)(t + 1) * DG_BM, HH) - 1));
        if (worst > DG_PPX) return SPAIR_ERR_UNSUPPORTED;
        tiles = B * tpi;
    }
    const bool stem = stem_part != nullptr;
    const int n_cu = spair_num_cus();
    const int grid = std::min(tiles, n_cu);      // persistent: one 160-KB workgroup per CU walks the tiles
    // one [128][17] partial per workgroup; the caller retries without the stem fusion when this is refused
    if (stem && (!stem_xp || !stem_dw || (long long)grid * DG_STEM_FLOATS > stem_part_cap || (stem_hin & 1) || (stem_s & 1))) return SPAIR_ERR_UNSUPPORTED;
    ConvDgradArgs a;
    a.dout = reinterpret_cast<const u16*>(dout);
    for (int q = 0; q < 4; ++q) a.Bz[q] = reinterpret_cast<const u16*>(wd[q]);
    a.gate = reinterpret_cast<const u16*>(gate); a.out = reinterpret_cast<u16*>(out);
    a.gbits = reinterpret_cast<const unsigned char*>(gate_bits);
    a.B = B; a.Ho = Ho; a.Hc = Hc; a.Hi = hin; a.M = (int)M; a.tpi = tpi;
    a.stem_xp = stem_xp; a.stem_part = stem_part; a.stem_hin = stem_hin; a.stem_s = stem_s;
    static std::atomic<unsigned long long> attr_done[4];
    const bool bits = gate_bits != nullptr;
    const int vi = (stem ? 1 : 0) + (bits ? 2 : 0);
    const void* fns[4] = {reinterpret_cast<const void*>(&k_conv_s2k4_dgrad<false, false>), reinterpret_cast<const void*>(&k_conv_s2k4_dgrad<true, false>),
                          reinterpret_cast<const void*>(&k_conv_s2k4_dgrad<false, true>), reinterpret_cast<const void*>(&k_conv_s2k4_dgrad<true, true>)};
    if (spair_dyn_lds_once(fns[vi], DG_LDS, attr_done[vi]) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    if (vi == 0) hipLaunchKernelGGL((k_conv_s2k4_dgrad<false, false>), dim3(grid), dim3(768), DG_LDS, s, a);
    else if (vi == 1) hipLaunchKernelGGL((k_conv_s2k4_dgrad<true, false>), dim3(grid), dim3(768), DG_LDS, s, a);
    else if (vi == 2) hipLaunchKernelGGL((k_conv_s2k4_dgrad<false, true>), dim3(grid), dim3(768), DG_LDS, s, a);
    else hipLaunchKernelGGL((k_conv_s2k4_dgrad<true, true>), dim3(grid), dim3(768), DG_LDS, s, a);
    SPAIR_CHECK_LAUNCH();
    if (stem) return spair_stem_fused_reduce(stem_part, grid, stem_dw, stem_db, s);
    return SPAIR_OK;
}

// unit-level C ABI (tests): wd4 = 4 device pointers, one per output-parity class
extern "C" int spair_conv_s2k4_dgrad16(const void* dout16, const void* wd0, const void* wd1, const void* wd2, const void* wd3, const void* gate16,
                                       void* out16, int B, int Ho, void* stream) {
    const void* wd[4] = {wd0, wd1, wd2, wd3};
    return conv_s2k4_patch_dgrad16(dout16, wd, gate16, out16, B, Ho, 2 * (Ho + 1), 128, 128, 4, 2, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                                   (hipStream_t)stream);
}
// the same with the ReLU gate given as sign bits (one byte per (pixel, 8 channels), [B][2 (Ho + 1)][2 (Ho + 1)][16]: spair_stem_conv_fwd_mask)
extern "C" int spair_conv_s2k4_dgrad16_bits(const void* dout16, const void* wd0, const void* wd1, const void* wd2, const void* wd3,
                                            const void* gate_bits8, void* out16, int B, int Ho, void* stream) {
    if (!gate_bits8) return SPAIR_ERR_SHAPE;
    const void* wd[4] = {wd0, wd1, wd2, wd3};
    return conv_s2k4_patch_dgrad16(dout16, wd, nullptr, out16, B, Ho, 2 * (Ho + 1), 128, 128, 4, 2, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                                   (hipStream_t)stream, gate_bits8);
}
