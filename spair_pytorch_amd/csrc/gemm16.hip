// bf16-STORED operand GEMMs (bf16 mode only): the backbone activations / gradients and the decoder's hidden
// activations live in HBM as bf16, so the big GEMMs move half the bytes of gemm.hip's fp32-operand kernels and
// stage 16-byte (8-element) chunks straight into LDS with no conversion.
//   gemm_nt16 : C[M,N] = epi(A[M,K] . B[N,K]^T), A bf16 row-major or an NHWC conv gather (ConvDesc), B bf16 weights,
//               C bf16 or fp32.  128x128x64 tiles, 4 waves (2x2, 64x64 each), LDS double-buffered: ONE barrier per
//               K-tile, the next tile's global loads are in flight during the 32 MFMAs of the current one.
//   gemm_tn16 : C[M,N] += sum_r A[r,M] . B[r,N] (weight gradients), A bf16, B bf16 (plain / conv gather) or fp32 gather
//               (first layer, Cin = image channels); operands staged row-major, fragments through ds_read_b64_tr_b16;
//               split over r with fp32 atomics; optional fused column sums of A (bias gradient); XCD-aware tile order.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "gemm.h"

struct ConvRow16 { int b, y, x; };
struct ConvTap16 { int ky, kx, ci; };

__device__ __forceinline__ void crow_init(const ConvDesc& c, int m, ConvRow16& r) {
    const int hw = c.Hout * c.Wout;
    r.b = m / hw;
    const int rem = m - r.b * hw;
    r.y = rem / c.Wout;
    r.x = rem - r.y * c.Wout;
}
// Branch-free (selects only): a data-dependent loop or `if` between a global load and its use makes hipcc's wait insertion
// fall back to s_waitcnt vmcnt(0) at the block boundaries, which serialises the software pipeline.  `wraps` = an upper bound
// on how many output rows one step can cross (ceil(step / Wout) -- computed on the host side of the kernel, wave-uniform).
__device__ __forceinline__ void crow_advance(const ConvDesc& c, ConvRow16& r, int step, int wraps) {
    r.x += step;
    for (int k = 0; k < wraps; ++k) {            // uniform trip count
        const bool w = r.x >= c.Wout;
        r.x -= w ? c.Wout : 0;
        r.y += w ? 1 : 0;
    }
    const bool h = r.y >= c.Hout;                // step < Hout*Wout: at most one image boundary per step
    r.y -= h ? c.Hout : 0;
    r.b += h ? 1 : 0;
}
__device__ __forceinline__ void ctap_init(const ConvDesc& c, int k, ConvTap16& t) {
    const int tap = k / c.Cin;
    t.ci = k - tap * c.Cin;
    t.ky = tap / c.kw;
    t.kx = tap - t.ky * c.kw;
}
__device__ __forceinline__ void ctap_advance(const ConvDesc& c, ConvTap16& t, int step, int wraps) {
    t.ci += step;
    for (int k = 0; k < wraps; ++k) {            // uniform trip count = ceil(step / Cin)
        const bool w = t.ci >= c.Cin;
        t.ci -= w ? c.Cin : 0;
        t.kx += w ? 1 : 0;
        const bool v = t.kx >= c.kw;
        t.kx = v ? 0 : t.kx;
        t.ky += v ? 1 : 0;
    }
}
// 8 consecutive k-elements of one tap (Cin % 8 == 0) of a bf16 NHWC tensor, zero outside
// `ok` folds the caller's row / k bounds; the load is always issued (clamped in-range address) and zeroed by a select
__device__ __forceinline__ uint4 conv_load8(const u16* __restrict__ In, const ConvDesc& c, const ConvRow16& r, const ConvTap16& t, bool ok) {
    const int sy = r.y * c.sy + c.oy + t.ky * c.dky;
    const int sx = r.x * c.sx + c.ox + t.kx * c.dkx;
    ok = ok && sy >= 0 && sy < c.Hin && sx >= 0 && sx < c.Win;
    const int cy = min(max(sy, 0), c.Hin - 1), cx = min(max(sx, 0), c.Win - 1), cc = min(t.ci, c.Cin - 8);
    const size_t off = (((size_t)r.b * c.Hin + cy) * c.Win + cx) * c.Cin + cc;
    const uint4 v = *reinterpret_cast<const uint4*>(In + off);
    return ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ float bf16_bits_to_float(u16 v) { return __uint_as_float(((unsigned int)v) << 16); }

typedef short v4s16_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_frag16(const __bf16* tile, int ld, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a0 = tile + (8 * g + q) * ld + c0 + 4 * p;
    const __bf16* a1 = a0 + 4 * ld;
    const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16_t*)(a0));
    const v4s16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16_t*)(a1));
    union { struct { v4s16_t lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

// STEM (conv_1's data gradient only): the stem's weight gradient is taken from the gated tile while it is still in LDS --
// dW0[ch][tap] = sum_rows G[row][ch] * patch(row)[tap] is one more 128x32x128 MFMA product per tile (column 16 of the patch
// matrix is 1: the bias gradient) -- and d act0 is never written to HBM (321 MB out + 321 MB back in at config 2).  Each
// workgroup leaves a [128][17] fp32 partial; spair_gemm_nt16_impl sums them (two small passes, no atomics).
#define STEM_PART_FLOATS (128 * 17)
// LDS operand tiles: rows of exactly BK elements (128 or 64 bytes) with the 16-byte chunk index XOR-ed by a function of the row
// -- chunk ^ (row & 7) at BK 64, chunk ^ (-(row >> 2) & 3) at BK 32 -- which makes every ds_read_b128 fragment read and every
// ds_write_b128 staging write conflict-free (ds_read_b128 serves lanes {0-3, 12-15, 20-27} etc. together: with rows padded by
// 16 bytes instead, 7 of each group's 16 lanes shared a bank pair with another, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32 - 0.46).
template <int BK>
__device__ __forceinline__ int nt16_swz(int row) { return BK == 64 ? (row & 7) : ((-(row >> 2)) & 3); }

// Staging: the operand tiles go HBM/L2 -> LDS by direct-to-LDS buffer loads (buffer_load_dwordx4 ... lds: no VGPR round trip, no
// ds_write pass; a wave-instruction writes 1 KiB = 8 (BK 64) or 16 (BK 32) whole unpadded tile rows at base + lane * 16, so the XOR
// swizzle is applied on the SOURCE side: the lane that owns LDS chunk c of row r fetches global chunk c ^ swz(r)); masked lanes pass an
// out-of-range offset and the range check writes zeros.
template <bool ACONV, bool C16, bool STEM, int BK>
__global__ __launch_bounds__(256, BK == 32 ? 3 : 2) void gemm_nt16_kernel(GemmNT g) {
    static_assert(BK == 64 || BK == 32, "K tile");
    constexpr int BM = 128, BN = 128, LD = BK;       // unpadded swizzled rows
    constexpr int WM = 64, WN = 64, TM = 4, TN = 4;
    constexpr int KQ = BK / 8, RPI = 256 / KQ;                          // 16-byte chunks per tile row; rows staged per pass of the block
    constexpr int NA = BM * KQ / 256, NB = BN * KQ / 256;               // 16-byte chunks per thread: 4 + 4 (BK 64), 2 + 2 (BK 32)
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    constexpr int NBUF = BK == 32 ? 3 : 2;      // BK 32: a 3-deep ring, two K tiles in flight behind a counted wait; BK 64: two buffers (73.7 KB, 2 workgroups per CU)
    __bf16* As0 = smem;                       // [NBUF][BM*LD]
    __bf16* Bs0 = smem + NBUF * BM * LD;      // [NBUF][BN*LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // parity-batched data gradients: the nz classes of one M tile read the same d-out rows (and the 2x2 taps overlap), so they run
    // back to back on ONE XCD (workgroup ids go round-robin over the 8 XCDs, each with its own L2): id -> (xcd, j), class = j % nz,
    // M tile = (j / nz) * 8 + xcd.  Class-major order (blockIdx.z) streamed d-out once per class: 773 MB read for a 76 MB tensor.
    int mt = blockIdx.x, zq = 0;
    if (g.nz > 1) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        zq = j % g.nz;
        mt = (j / g.nz) * 8 + xcd;
        if (mt * BM >= g.M) return;                  // grid padded to a whole number of (8 XCD x nz) groups
    }
    // several N tiles: the tiles of one M block read the same A rows, so they too run back to back on one XCD (column-tile-major
    // order re-streamed A once per N tile: the decoder.out data gradient read its 205 MB of d-logits twice and was HBM bound)
    int ntile = blockIdx.y;
    if (g.xcd_tiles_n > 1) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        ntile = j % g.xcd_tiles_n;
        mt = (j / g.xcd_tiles_n) * 8 + xcd;
        if (mt * BM >= g.M) return;
    }
    const int m0 = mt * BM, n0 = ntile * BN;
    const u16* A = reinterpret_cast<const u16*>(g.A);
    const u16* B = reinterpret_cast<const u16*>(g.nz > 1 ? g.Bz[zq] : g.B);
    const int cm_ooy = g.nz > 1 ? zq / g.cmap.osx : g.cmap.ooy, cm_oox = g.nz > 1 ? zq - (zq / g.cmap.osx) * g.cmap.osx : g.cmap.oox;

    // staging coordinates: chunk f = tid + i*256 -> row f/8, k-chunk f%8 (the same k-chunk for all of a thread's chunks).
    // Address arithmetic is the bottleneck of a conv gather if done naively (a 64-bit multiply chain per 16-byte chunk made both
    // conv kernels VALU-issue bound): every row's element offset is computed ONCE (32-bit), the tap's offset once per K tile, and a
    // chunk's address is one add.  Tensors must stay below 2^31 elements (checked by the launcher).
    const int kq_slot = tid % KQ;                                        // this thread's LDS chunk slot in its rows
    const int kq = kq_slot ^ nt16_swz<BK>(tid / KQ);                    // ... and the global k-chunk it fetches (source-side swizzle)
    ConvTap16 a_ct;
    unsigned a_base[NA];                 // element offset of (b, y*sy+oy, x*sx+ox, 0) (conv) or of row m (plain)
    int a_y[NA], a_x[NA];                // conv: y*sy+oy, x*sx+ox for the bounds test of data-gradient gathers
    bool a_ok[NA], b_ok[NB];
    unsigned b_base[NB];
    a_ct.ky = a_ct.kx = a_ct.ci = 0;
    if (ACONV) ctap_init(g.conv, kq * 8, a_ct);
    const bool need_bounds = ACONV && (g.conv.dky < 0 || g.conv.oy != 0 || g.conv.ox != 0 || g.conv.dkx < 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = tid / KQ + i * RPI;
        a_ok[i] = (m0 + row) < g.M;
        const int mr = min(m0 + row, g.M - 1);
        a_y[i] = a_x[i] = 0;
        if (ACONV) {
            ConvRow16 cr;
            crow_init(g.conv, mr, cr);
            a_y[i] = cr.y * g.conv.sy + g.conv.oy;
            a_x[i] = cr.x * g.conv.sx + g.conv.ox;
            a_base[i] = (unsigned)(((cr.b * g.conv.Hin + a_y[i]) * g.conv.Win + a_x[i]) * g.conv.Cin);   // may wrap for negative taps: fixed by tapoff
        } else {
            a_base[i] = (unsigned)mr * (unsigned)g.lda;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = tid / KQ + i * RPI;
        b_ok[i] = (n0 + row) < g.N;
        b_base[i] = (unsigned)min(n0 + row, g.N - 1) * (unsigned)g.ldb;
    }

    const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(A), rsB = buf_rsrc(B);
    const int tap_wraps = ACONV ? ceil_div_dev(BK, g.conv.Cin) : 0;
    const int Klast = g.K - 8;
    // the loads land in LDS buffer `buf` directly (no branch around a load: a conditional VM op makes the pending count unknown and every
    // wait becomes vmcnt(0); tiles past K are requested out of range and land as zeros)
    auto glds16 = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte_off, __bf16* tile, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(tile) + (i * 256 + wave * 64) * 16), 16,
                                                 (int)byte_off, 0, 0, 0);
    };
    auto glds_tiles = [&](int k0, int buf) {
        __bf16* As = As0 + buf * BM * LD;
        __bf16* Bs = Bs0 + buf * BN * LD;
        const int k = k0 + kq * 8;
        const bool kok = k < g.K;
        const bool live = k0 < g.K;                      // wave-uniform
        const unsigned kc = (unsigned)min(k, Klast);
        if (ACONV) {
            int t_ky = a_ct.ky, t_kx = a_ct.kx, t_ci = a_ct.ci;
            if (BK == 64 && g.n_ktab > 0) {
                const unsigned e = g.ktab[min(k0 >> 6, g.n_ktab - 1)];
                t_ky = (int)(e >> 24); t_kx = (int)((e >> 16) & 255u); t_ci = (int)(e & 0xffffu) + kq * 8;
            }
            const int dy = t_ky * g.conv.dky, dx = t_kx * g.conv.dkx;
            const unsigned tapoff = (unsigned)((dy * g.conv.Win + dx) * g.conv.Cin + min(t_ci, g.conv.Cin - 8));
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                bool ok = a_ok[i] && kok;
                unsigned off = a_base[i] + tapoff;
                if (need_bounds) {
                    const int sy = a_y[i] + dy, sx = a_x[i] + dx;
                    const bool in = sy >= 0 && sy < g.conv.Hin && sx >= 0 && sx < g.conv.Win;
                    ok = ok && in;
                    off = in ? off : 0u;
                }
                glds16(rsA, (ok && live) ? off * 2u : BUF_OOB, As, i);
            }
            if (live) ctap_advance(g.conv, a_ct, BK, tap_wraps);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(rsA, (a_ok[i] && kok) ? (a_base[i] + kc) * 2u : BUF_OOB, As, i);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(rsB, (b_ok[i] && kok) ? (b_base[i] + kc) * 2u : BUF_OOB, Bs, i);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (g.K + BK - 1) / BK;
    auto mfma_tile = [&](int buf) {
        const __bf16* As = As0 + buf * BM * LD;
        const __bf16* Bs = Bs0 + buf * BN * LD;
        const int arow = wm * WM + (lane & 15), brow = wn * WN + (lane & 15);
        const int sw = nt16_swz<BK>(lane & 15);                 // the fragment rows differ from lane & 15 by multiples of 16
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[TM], bfr[TN];
            const int kg = ((ks * 4 + (lane >> 4)) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + i * 16) * LD + kg]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + j * 16) * LD + kg]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NBUF == 3) {
        // ring of three: tile kt+2 is issued before the MFMAs of tile kt; the wait before the (raw) barrier only retires tile kt+1's
        // NA + NB loads -- tile kt+2 stays in flight across it (a __syncthreads() would drain it: it waits vmcnt(0) with LDS-DMA pending).
        // Buffer (kt+2) % 3 was last read in step kt-1, behind that step's barrier.
        glds_tiles(0, 0);
        glds_tiles(BK, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
        __builtin_amdgcn_s_barrier();
        int bcur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int bnext2 = bcur == 0 ? 2 : bcur - 1;       // (kt + 2) % 3
            glds_tiles((kt + 2) * BK, bnext2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(bcur);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
            __builtin_amdgcn_s_barrier();
            bcur = bcur == 2 ? 0 : bcur + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the epilogue re-uses the operand buffers
        __syncthreads();
    } else {
        glds_tiles(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            glds_tiles((kt + 1) * BK, (kt + 1) & 1);     // the other buffer: last read one iteration ago, behind the previous barrier
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(kt & 1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // epilogue.  The accumulators go through LDS (the operand buffers are free now) so that everything that touches HBM is a
    // coalesced 16-byte-per-lane access of whole rows: the relu-mask read (data gradients), the row-remapped store, the sprite
    // sigmoid.  Straight from the MFMA layout a lane would issue 64 scattered 2-byte mask loads and 64 2/4-byte stores of
    // 32/64-byte segments (decoder.out: 411 MB of fp32 sprites left at 1 TB/s).
    constexpr int LDC = BN + 4;                       // fp32 staging tile [BM][LDC]: 67.6 KB of the 73.7 KB (BK 64)
    constexpr int NH = BK == 32 ? 2 : 1;              // BK 32 (40 KB of LDS, 3 workgroups per CU): the tile is staged in two 64-row passes
    constexpr int RH = BM / NH;
    float* Cs = reinterpret_cast<float*>(smem);
    const bool vec_ok = C16 ? ((g.ldc & 7) == 0) : ((g.ldc & 3) == 0);
    const int c8 = (tid & 15) * 8;                    // this thread's 8 columns; rows (tid>>4) + 16*i
    const int nb = n0 + c8;
    const bool full = (nb + 8) <= g.N;
    // destination rows and (data gradients) the relu-gate chunks of all 8 rows first: the gate loads are in flight while the
    // accumulators are staged -- issued one per row inside the store loop they cost 8 serialised HBM round trips per tile
    // (23 us of fixed cost on a tile whose 8 K-steps take 8 us)
    constexpr int NR = BM / 16;
    size_t crow[NR];
    uint4 gate[NR];
    const bool vec_gate = g.mask != nullptr && g.mask_bf16 && (g.ldmask & 7) == 0 && full;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int m = min(m0 + (tid >> 4) + i * 16, g.M - 1);
        if (g.use_cmap) {
            const int hw = g.cmap.Hout * g.cmap.Wout;
            const int b = m / hw, rem = m - b * hw, y = rem / g.cmap.Wout, x = rem - y * g.cmap.Wout;
            crow[i] = ((size_t)b * g.cmap.Hc + (y * g.cmap.osy + cm_ooy)) * g.cmap.Wc + (x * g.cmap.osx + cm_oox);
        } else {
            crow[i] = (size_t)m;
        }
        gate[i] = make_uint4(0u, 0u, 0u, 0u);
        if (vec_gate) gate[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(g.mask) + crow[i] * g.ldmask + nb);
    }
    // STEM: this thread's half patch (row tid>>1, input rows ky = 2*half, 2*half+1; 4 taps each), in flight during the staging
    float2 sp[4];
    bool sp_ok = false;
    if (STEM) {
        const int prow = tid >> 1, half = tid & 1;
        const int m = m0 + prow;
        sp_ok = m < g.M;
        const int mc = min(m, g.M - 1);
        const int hw = g.cmap.Hout * g.cmap.Wout;
        const int b = mc / hw, rem = mc - b * hw, y = rem / g.cmap.Wout, x = rem - y * g.cmap.Wout;
        const int yy = y * g.cmap.osy + cm_ooy, xx = x * g.cmap.osx + cm_oox;
        const float* src = g.stem_xp + ((size_t)b * g.stem_hin + yy * g.stem_s + 2 * half) * g.stem_hin + xx * g.stem_s;
        sp[0] = *reinterpret_cast<const float2*>(src);
        sp[1] = *reinterpret_cast<const float2*>(src + 2);
        sp[2] = *reinterpret_cast<const float2*>(src + g.stem_hin);
        sp[3] = *reinterpret_cast<const float2*>(src + g.stem_hin + 2);
    }
    uint4 gq[STEM ? NR : 1];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
    if (hh > 0) __syncthreads();                      // the previous pass has been read
    if (NH == 1 || wm == hh) {
        const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WN + j * 16 + col_l;
            const float bv = (g.bias && (n0 + nl) < g.N) ? g.bias[n0 + nl] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(wm * WM + i * 16 + rgrp + r - hh * RH) * LDC + nl] = acc[i][j][r] + bv;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < NR / NH; ++ii) {
        const int i = hh * (NR / NH) + ii;
        const int rl = (tid >> 4) + i * 16;
        const int m = m0 + rl;
        if (STEM) gq[i] = make_uint4(0u, 0u, 0u, 0u);
        if (m >= g.M || nb >= g.N) continue;
        float v[8];
        {
            const float4 q0 = *reinterpret_cast<const float4*>(&Cs[(rl - hh * RH) * LDC + c8]);
            const float4 q1 = *reinterpret_cast<const float4*>(&Cs[(rl - hh * RH) * LDC + c8 + 4]);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        }
        if (g.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (g.mask) {
            if (vec_gate) {
                const unsigned mw[4] = {gate[i].x, gate[i].y, gate[i].z, gate[i].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned h = (mw[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                    v[e] = ((h & 0x8000u) == 0 && (h & 0x7fffu) != 0) ? v[e] : 0.f;          // bf16 value > 0
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (nb + e >= g.N) continue;
                    bool on;
                    if (g.mask_bf16) {
                        const u16 h = reinterpret_cast<const u16*>(g.mask)[crow[i] * g.ldmask + nb + e];
                        on = (h & 0x8000u) == 0 && (h & 0x7fffu) != 0;
                    } else {
                        on = g.mask[crow[i] * g.ldmask + nb + e] > 0.f;
                    }
                    v[e] = on ? v[e] : 0.f;
                }
            }
        }
        if (C16 && g.sprite_ch == 2) {
            // (grey, alpha) pairs, nb is a multiple of 8: even columns are grey, odd ones alpha -- the scales are per-lane constants,
            // log2(e) is folded into them (v_exp_f32 is 2^x) and the FMA / add run packed: 4 VALU + 4 transcendental issues per
            // column pair instead of ~20 (the epilogue cost 70 us on top of the 120 us plain GEMM)
            typedef float f2 __attribute__((ext_vector_type(2)));
            constexpr float L2E = 1.4426950408889634f;
            const f2 sc = f2{-g.obj_scale * L2E, -g.alpha_scale * L2E}, bi = f2{0.f, -g.alpha_bias * L2E}, one = f2{1.f, 1.f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f2 u = __builtin_elementwise_fma(f2{v[e], v[e + 1]}, sc, bi);
                const f2 d = f2{__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)} + one;
                v[e] = __builtin_amdgcn_rcpf(d.x);
                v[e + 1] = __builtin_amdgcn_rcpf(d.y);
            }
        } else if (g.sprite_ch > 0) {
            int ch = nb % g.sprite_ch;                // one modulo per chunk, then a running channel index
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (ch == g.sprite_ch - 1) ? v[e] * g.alpha_scale + g.alpha_bias : v[e] * g.obj_scale;
                // bf16 sprites carry 8 mantissa bits: the hardware exp2 / rcp (1 ulp) are exact enough and 10x cheaper than expf + IEEE divide
                v[e] = C16 ? __builtin_amdgcn_rcpf(__expf(-t) + 1.f) : 1.f / (expf(-t) + 1.f);
                ch = (ch + 1 == g.sprite_ch) ? 0 : ch + 1;
            }
        }
        if (STEM) {          // kept for the fused weight-gradient product below; nothing goes to HBM
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((nb + e) < g.N ? v[e] : 0.f);
            gq[i] = *reinterpret_cast<uint4*>(&o);
        } else if (C16 && g.sprite_ch > 0) {
            // sprites are sigmoid outputs in (0, 1): stored as FP16 (11 significant bits, no range problem) rather than bf16 (8) -- the
            // same bytes, and the reconstruction / ELBO error of the bf16 step drops with it (the renderer unpacks either in one instruction)
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            _Float16* dst = reinterpret_cast<_Float16*>(g.C) + crow[i] * g.ldc + nb;
            if (full && vec_ok) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
                *reinterpret_cast<h8*>(dst) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (nb + e < g.N) dst[e] = (_Float16)v[e];
            }
        } else if (C16) {
            __bf16* dst = reinterpret_cast<__bf16*>(g.C) + crow[i] * g.ldc + nb;
            if (full && vec_ok) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x8*>(dst) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (nb + e < g.N) dst[e] = (__bf16)v[e];
            }
        } else {
            float* dst = g.C + crow[i] * g.ldc + nb;
            if (full && vec_ok) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (nb + e < g.N) dst[e] = v[e];
            }
        }
    }
    }
    if (STEM) {
        constexpr int LDG = BN + 8, LDP = 32 + 8;
        __bf16* Gs = smem;                                   // [BM rows][LDG]: the gated tile, bf16 (what d act0 would have held)
        __bf16* Ps = smem + BM * LDG;                        // [BM rows][LDP]: 16 taps | 1 | 0 ...
        float* Ws = reinterpret_cast<float*>(smem);          // [128 ch][17]: over the gated tile once the MFMAs have read it (45 KB in all)
        __syncthreads();                                     // every thread has read its part of Cs
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<uint4*>(&Gs[((tid >> 4) + i * 16) * LDG + c8]) = gq[i];
        {
            const int prow = tid >> 1, half = tid & 1;
            bf16x8 o, one;
            const float pv[8] = {sp[0].x, sp[0].y, sp[1].x, sp[1].y, sp[2].x, sp[2].y, sp[3].x, sp[3].y};
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = (__bf16)(sp_ok ? pv[e] : 0.f); one[e] = (__bf16)0.f; }
            one[0] = (__bf16)((sp_ok && half == 0) ? 1.f : 0.f);
            *reinterpret_cast<bf16x8*>(&Ps[prow * LDP + half * 8]) = o;
            *reinterpret_cast<bf16x8*>(&Ps[prow * LDP + 16 + half * 8]) = one;
        }
        __syncthreads();
        f32x4 sacc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) sacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < BM / 32; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = lds_tr_frag16(Gs + ks * 32 * LDG, LDG, wave * 32 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = lds_tr_frag16(Ps + ks * 32 * LDP, LDP, j * 16, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) sacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], sacc[i][j], 0, 0, 0);
        }
        __syncthreads();                                     // Gs is free
        {
            const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = wave * 32 + i * 16 + rgrp + r;
                    Ws[ch * 17 + col_l] = sacc[i][0][r];
                    if (col_l == 0) Ws[ch * 17 + 16] = sacc[i][1][r];
                }
        }
        __syncthreads();
        float4* pt = reinterpret_cast<float4*>(g.stem_part + ((size_t)zq * ceil_div_dev(g.M, BM) + mt) * STEM_PART_FLOATS);
        for (int q = tid; q < STEM_PART_FLOATS / 4; q += 256) pt[q] = reinterpret_cast<const float4*>(Ws)[q];
    }
}

// sums of the fused stem partials: part[nblk][128*17] -> part2[S][128*17] -> dW0 / db0
__global__ __launch_bounds__(256) void k_stem_fused_reduce1(const float* __restrict__ part, int nblk, int per, float* __restrict__ part2) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= STEM_PART_FLOATS) return;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = part[(size_t)(b + e) * STEM_PART_FLOATS + col];
        t0 += v[0] + v[4]; t1 += v[1] + v[5]; t2 += v[2] + v[6]; t3 += v[3] + v[7];
    }
    for (; b < b1; ++b) t0 += part[(size_t)b * STEM_PART_FLOATS + col];
    part2[(size_t)blockIdx.y * STEM_PART_FLOATS + col] = (t0 + t1) + (t2 + t3);
}
__global__ __launch_bounds__(256) void k_stem_fused_reduce2(const float* __restrict__ part2, int S, float* __restrict__ dW, float* __restrict__ db) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= STEM_PART_FLOATS) return;
    float t = 0.f;
    for (int q0 = 0; q0 < S; q0 += 8) {      // 8 independent loads in flight (S is a multiple of 8)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = part2[(size_t)(q0 + e) * STEM_PART_FLOATS + col];
#pragma unroll
        for (int e = 0; e < 8; ++e) t += v[e];
    }
    const int ch = col / 17, n = col - ch * 17;
    if (n < 16) dW[ch * 16 + n] += t;
    else if (db) db[ch] += t;
}
// part[nblk][128*17] (+ 64 * 128*17 floats of scratch behind it) -> dw [128][16] += , db [128] +=   (two small deterministic passes)
int spair_stem_fused_reduce(float* part, int nblk, float* dw, float* db, hipStream_t s) {
    const int S = 64, per = ceil_div(nblk, S);
    float* part2 = part + (size_t)nblk * STEM_PART_FLOATS;
    hipLaunchKernelGGL(k_stem_fused_reduce1, dim3(ceil_div(STEM_PART_FLOATS, 256), S), dim3(256), 0, s, part, nblk, per, part2);
    SPAIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_stem_fused_reduce2, dim3(ceil_div(STEM_PART_FLOATS, 256)), dim3(256), 0, s, part2, S, dw, db);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
bool spair_nt16_stem_fusable(const GemmNT& g, long long part_cap) {
    const long long tiles = (long long)ceil_div(g.M, 128) * (g.nz > 1 ? g.nz : 1);
    return g.N == 128 && g.use_cmap && g.c_bf16 && (g.stem_hin % 2) == 0 && (g.stem_s % 2) == 0 &&
           (tiles + 64) * STEM_PART_FLOATS <= part_cap;
}

int spair_gemm_nt16_impl(const GemmNT& g_in, bool conv, hipStream_t s) {
    GemmNT g = g_in;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return SPAIR_ERR_SHAPE;
    if ((g.K & 7) || (!conv && (g.lda & 7)) || (g.ldb & 7)) return SPAIR_ERR_ALIGN;
    if (conv && (g.conv.Cin & 7)) return SPAIR_ERR_ALIGN;
    if (g.accumulate) return SPAIR_ERR_UNSUPPORTED;
    if (!conv && (long long)g.M * g.lda >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;   // 32-bit element offsets
    if ((long long)g.N * g.ldb >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    if (conv) {      // the gathered operand and the (possibly row-mapped) output are addressed with 32-bit element offsets too
        const long long rows_in = (long long)ceil_div(g.M, g.conv.Hout * g.conv.Wout) * g.conv.Hin * g.conv.Win;
        if (rows_in * g.conv.Cin >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    }
    {
        const long long rows_out = g.use_cmap ? (long long)ceil_div(g.M, g.cmap.Hout * g.cmap.Wout) * g.cmap.Hc * g.cmap.Wc : (long long)g.M;
        if (rows_out * g.ldc >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    }
    // K tile: 64 (two operand buffers, 73.7 KB of LDS with the epilogue staging tile, 2 workgroups per CU) or 32 (ring of three, 48 KB, 3 per
    // CU, the epilogue staged in two passes).  Measured: the long-K launches are faster at 64 (conv_1 forward 0.263 vs 0.288 ms, decoder.out
    // data gradient 0.099 vs 0.128 ms), the short-K ones, where the epilogue weighs most, at 32 (decoder.out forward, K = 256: 0.172 -> 0.148).
    if (g.n_ktab > 0 && (!conv || g.n_ktab > 64 || g.n_ktab * 64 != g.K || (g.conv.Cin & 63))) return SPAIR_ERR_SHAPE;
    const int bk = (g.n_ktab > 0 || g.K >= 1024) ? 64 : 32;
    const int nbuf = bk == 32 ? 3 : 2;
    size_t lds = std::max((size_t)(2 * nbuf) * 128 * bk * 2, (size_t)(bk == 32 ? 64 : 128) * (128 + 4) * 4);   // operands | epilogue staging
    if (g.stem_part) lds = std::max(lds, (size_t)128 * (128 + 8 + 32 + 8) * 2);      // gated tile + patches, bf16
    if (g.nz > 1 && (g.nz > 4 || !g.use_cmap || g.nz != g.cmap.osy * g.cmap.osx)) return SPAIR_ERR_SHAPE;
    dim3 grid(ceil_div(g.M, 128), ceil_div(g.N, 128), 1);
    if (g.nz > 1) grid.x = (unsigned)(ceil_div(ceil_div(g.M, 128), 8) * 8 * g.nz);      // (8 XCDs) x (nz classes) x ceil(tiles / 8)
    g.xcd_tiles_n = 0;
    if (g.nz <= 1 && grid.y > 1) {
        g.xcd_tiles_n = (int)grid.y;
        grid.x = (unsigned)(ceil_div(ceil_div(g.M, 128), 8) * 8 * g.xcd_tiles_n);
        grid.y = 1;
    }
#define NT16_LAUNCH_BK(AC, C16, ST, BKV)                                                                        \
    do {                                                                                                          \
        static std::atomic<unsigned long long> attr_done{0};                                                      \
        spair_dyn_lds_once(reinterpret_cast<const void*>(&gemm_nt16_kernel<AC, C16, ST, BKV>), (int)lds, attr_done); \
        hipLaunchKernelGGL((gemm_nt16_kernel<AC, C16, ST, BKV>), grid, dim3(256), lds, s, g);                     \
    } while (0)
#define NT16_LAUNCH(AC, C16, ST)                                                                                 \
    do { if (bk == 32) NT16_LAUNCH_BK(AC, C16, ST, 32); else NT16_LAUNCH_BK(AC, C16, ST, 64); } while (0)
    if (g.stem_part) {   // conv_1's data gradient with the stem's weight gradient taken in the epilogue
        if (!conv || !g.stem_xp || !g.stem_dw || !spair_nt16_stem_fusable(g, g.stem_part_cap)) return SPAIR_ERR_UNSUPPORTED;
        NT16_LAUNCH(true, true, true);
        SPAIR_CHECK_LAUNCH();
        return spair_stem_fused_reduce(g.stem_part, ceil_div(g.M, 128) * (g.nz > 1 ? g.nz : 1), g.stem_dw, g.stem_db, s);
    }
    if (conv) { if (g.c_bf16) NT16_LAUNCH(true, true, false); else NT16_LAUNCH(true, false, false); }
    else { if (g.c_bf16) NT16_LAUNCH(false, true, false); else NT16_LAUNCH(false, false, false); }
#undef NT16_LAUNCH
#undef NT16_LAUNCH_BK
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---------------------------------------------------------------------------------------------
// TN (weight gradients) with bf16-stored A (and B)
// ---------------------------------------------------------------------------------------------
// B16: B is bf16 (plain rows or conv gather with Cin % 8 == 0); otherwise B is an fp32 conv gather (any Cin, scalar loads)
template <bool BCONV, bool B16>
__global__ __launch_bounds__(256, 2) void gemm_tn16_kernel(GemmTN g) {
    constexpr int BM = 128, BN = 128, BK = 32, LDA = BM + 8, LDB = BN + 8;
    constexpr int WM = 64, WN = 64, TM = 4, TN = 4;
    constexpr int NA = BK * (BM / 8) / 256;        // 2 chunks of 8 per thread
    constexpr int NB16 = BK * (BN / 8) / 256;      // 2
    constexpr int NB32 = BK * (BN / 4) / 256;      // 4 (fp32 gather, 4 elements per chunk)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int mt_, nt_, sp_;
    {
        const int id = blockIdx.x, ntm = g.tiles_m, ntn = g.tiles_n;
        if ((g.nsplit & 7) == 0) {   // XCD-aware: all tiles of one row split on one XCD (see gemm.hip)
            const int xcd = id & 7, j = id >> 3;
            mt_ = j % ntm; nt_ = (j / ntm) % ntn; sp_ = (j / (ntm * ntn)) * 8 + xcd;
        } else {
            mt_ = id % ntm; nt_ = (id / ntm) % ntn; sp_ = id / (ntm * ntn);
        }
    }
    // grouped launch (ngroup > 1): the n-tile index selects one of the independent single-tile problems of g.tile[] (a table in the
    // kernel-argument segment: the index is wave-uniform, the entries arrive through scalar loads and stay global pointers)
    const bool grouped = g.ngroup > 1;
    const GemmTN::Tile& gt = g.tile[grouped ? nt_ : 0];
    const float* gA = grouped ? reinterpret_cast<const float*>(gt.A) : g.A;
    const float* gB = grouped ? reinterpret_cast<const float*>(gt.B) : g.B;
    const int g_lda = grouped ? gt.lda : g.lda;
    const int g_ldb = grouped ? gt.ldb : g.ldb;
    const int g_M = grouped ? gt.M : g.M;
    const int g_N = grouped ? gt.N : g.N;
    const int g_Mstore = grouped ? gt.Mstore : g.Mstore;
    const int g_mskip = grouped ? gt.m_skip : 0;
    float* g_colsum = grouped ? gt.colsum : g.colsum_out;
    const int m0 = mt_ * BM, n0 = grouped ? 0 : nt_ * BN;
    const int r_begin = sp_ * g.rows_per_split;
    const int r_end = min(g.R, r_begin + g.rows_per_split);
    const u16* A = reinterpret_cast<const u16*>(gA);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A chunks: f = tid + i*256 -> row kr = f / 16, column chunk mq = f % 16 (fixed per thread)
    const int a_mq = tid & 15;
    uint4 ra[NA];
    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    const bool do_colsum = g_colsum != nullptr && (grouped || nt_ == 0);
    // B chunks
    constexpr int NB = B16 ? NB16 : NB32;
    uint4 rb16[NB16];
    float4 rb32[NB32];
    // Running 32-bit element offsets (see gemm_nt16): A advances by BK rows per tile; a conv-gathered B row (b,y,x) advances by BK
    // output pixels with branch-free row / image wraps whose offset deltas are wave-uniform constants.
    int b_x[NB], b_y[NB];
    unsigned b_off[NB];
    int b_tapoff[NB][4];
    bool b_nok[NB];
    bool b_vec = true;
    unsigned a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_off[i] = (unsigned)(r_begin + (tid >> 4) + i * 16) * (unsigned)g_lda + (unsigned)min(m0 + a_mq * 8, g_M - 8);
    const bool a_mok = (m0 + a_mq * 8) < g_M;
    const unsigned a_last = (unsigned)(g.R - 1) * (unsigned)g_lda + (unsigned)min(m0 + a_mq * 8, g_M - 8);
    const int row_wraps = BCONV ? ceil_div_dev(BK, g.conv.Wout) : 0;
    const int d_step = BCONV ? BK * g.conv.sx * g.conv.Cin : BK * g_ldb;                                  // x += BK
    const int d_row = BCONV ? (g.conv.sy * g.conv.Win - g.conv.Wout * g.conv.sx) * g.conv.Cin : 0;      // x -= Wout, y += 1
    const int d_img = BCONV ? (g.conv.Hin - g.conv.Hout * g.conv.sy) * g.conv.Win * g.conv.Cin : 0;     // y -= Hout, b += 1
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = tid + i * 256;
        const int kr = B16 ? f / (BN / 8) : f / (BN / 4);
        const int nq = B16 ? f % (BN / 8) : f % (BN / 4);
        const int n = n0 + nq * (B16 ? 8 : 4);
        b_nok[i] = n < g_N;
        b_x[i] = b_y[i] = 0;
        if (BCONV) {
            b_vec = B16 ? true : (g.conv.Cin & 3) == 0;
            ConvRow16 cr;
            crow_init(g.conv, min(r_begin + kr, g.R - 1), cr);
            b_x[i] = cr.x; b_y[i] = cr.y;
            b_off[i] = (unsigned)(((cr.b * g.conv.Hin + cr.y * g.conv.sy + g.conv.oy) * g.conv.Win + cr.x * g.conv.sx + g.conv.ox) * g.conv.Cin);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ConvTap16 t;
                ctap_init(g.conv, min(n + e, g_N - 1), t);
                b_tapoff[i][e] = (t.ky * g.conv.dky * g.conv.Win + t.kx * g.conv.dkx) * g.conv.Cin + t.ci;
            }
        } else {
            b_off[i] = (unsigned)(r_begin + kr) * (unsigned)g_ldb + (unsigned)min(n, g_N - (B16 ? 8 : 4));
            b_tapoff[i][0] = b_tapoff[i][1] = b_tapoff[i][2] = b_tapoff[i][3] = 0;
        }
    }
    const unsigned b_lastrow = BCONV ? 0u : (unsigned)(g.R - 1) * (unsigned)g_ldb;
    const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(A), rsB = buf_rsrc(gB);
    auto load_tiles = [&](int r0) {      // every load is issued unconditionally from a clamped address and zeroed by a select
#pragma unroll
        for (int i = 0; i < NA; ++i) {       // buffer loads: a masked lane goes out of range (zeros), nothing is conditional (see buf_load16)
            const int r = r0 + (tid >> 4) + i * 16;
            ra[i] = buf_load16(rsA, (r < r_end && a_mok) ? min(a_off[i], a_last) * 2u : BUF_OOB);
            a_off[i] += (unsigned)(BK * g_lda);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            const int kr = B16 ? f / (BN / 8) : f / (BN / 4);
            const bool ok = (r0 + kr) < r_end && b_nok[i];
            // rows past the end of the tensor (last split only) are clamped by construction for conv (crow_init clamps the start and
            // the row test zeroes the value; offsets only ever move forward inside the allocation of the last image) and explicitly here
            unsigned off = b_off[i];
            if (!BCONV) off = min(off, b_lastrow + (unsigned)(g_ldb - (B16 ? 8 : 4)));
            if (BCONV && !ok) off = 0u;
            if constexpr (B16) {
                rb16[i] = buf_load16(rsB, ok ? (off + (unsigned)b_tapoff[i][0]) * 2u : BUF_OOB);
            } else {
                const float* Bp = gB;
                float4 v;
                if (b_vec) v = *reinterpret_cast<const float4*>(Bp + off + (unsigned)b_tapoff[i][0]);     // wave-uniform choice
                else v = make_float4(Bp[off + (unsigned)b_tapoff[i][0]], Bp[off + (unsigned)b_tapoff[i][1]], Bp[off + (unsigned)b_tapoff[i][2]],
                                     Bp[off + (unsigned)b_tapoff[i][3]]);
                rb32[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // advance to the row BK further on
            b_off[i] += (unsigned)d_step;
            if (BCONV) {
                b_x[i] += BK;
                for (int kk = 0; kk < row_wraps; ++kk) {        // uniform trip count
                    const bool w = b_x[i] >= g.conv.Wout;
                    b_x[i] -= w ? g.conv.Wout : 0;
                    b_y[i] += w ? 1 : 0;
                    b_off[i] += w ? (unsigned)d_row : 0u;
                }
                const bool h = b_y[i] >= g.conv.Hout;
                b_y[i] -= h ? g.conv.Hout : 0;
                b_off[i] += h ? (unsigned)d_img : 0u;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            // opaque copies: whatever consumes the loaded registers is tied below the scheduling barrier that follows the MFMAs
            // (otherwise the bf16 unpacking of the column sums is hoisted above them together with an s_waitcnt vmcnt(0))
            uint4 av = ra[i];
            asm volatile("" : "+v"(av.x), "+v"(av.y), "+v"(av.z), "+v"(av.w));
            *reinterpret_cast<uint4*>(&As[buf][((tid >> 4) + i * 16) * LDA + a_mq * 8]) = av;
            if (do_colsum) {
                const u16* h = reinterpret_cast<const u16*>(&av);
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[e] += bf16_bits_to_float(h[e]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = tid + i * 256;
            if constexpr (B16) {
                const int kr = f / (BN / 8), nq = f % (BN / 8);
                *reinterpret_cast<uint4*>(&Bs[buf][kr * LDB + nq * 8]) = rb16[i];
            } else {
                const int kr = f / (BN / 4), nq = f % (BN / 4);
                bf16x4 v;
                v[0] = (__bf16)rb32[i].x; v[1] = (__bf16)rb32[i].y; v[2] = (__bf16)rb32[i].z; v[3] = (__bf16)rb32[i].w;
                *reinterpret_cast<bf16x4*>(&Bs[buf][kr * LDB + nq * 4]) = v;
            }
        }
    };

    if (r_begin < r_end) {
        load_tiles(r_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int r0 = r_begin; r0 < r_end; r0 += BK) {
            const bool more = (r0 + BK) < r_end;
            if (more) load_tiles(r0 + BK);
            // pin the schedule: loads are ISSUED above, consumed (column sums, LDS stores) only below the MFMAs.  Without the two
            // scheduling barriers hipcc hoists the bf16->fp32 column-sum math of the freshly loaded registers above the MFMAs and
            // with it an s_waitcnt vmcnt(0): the whole load latency is then exposed in every iteration.
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = lds_tr_frag16(As[buf], LDA, wm * WM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = lds_tr_frag16(Bs[buf], LDB, wn * WN + j * 16, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    if (do_colsum) {   // bias gradient: column sums of A over this block's rows, reduced in LDS, one atomic per column
        __syncthreads();
        float* scr = reinterpret_cast<float*>(&As[0][0]);      // 16 row groups x 128 columns of fp32 = 8 KB
        const int grp = tid >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) scr[grp * BM + a_mq * 8 + e] = csum[e];
        __syncthreads();
        for (int m = tid; m < BM; m += 256) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += scr[q * BM + m];
            // split-K partial path: this block's column sums go to colpart[split][tile][BM] and k_tn_reduce adds the splits in a fixed
            // order (run-to-run identical bias gradients); without scratch, one atomic per column
            if (g.colpart) g.colpart[((size_t)sp_ * (g.tiles_m * g.tiles_n) + (size_t)nt_ * g.tiles_m + mt_) * BM + m] = t;
            else if (m0 + m >= g_mskip && m0 + m < g_mskip + g_Mstore) atomicAdd(&g_colsum[m0 + m - g_mskip], t);
        }
    }
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
    if (g.part) {
        // split-K partial: plain stores, 64 B per 16 lanes; summed by k_tn_reduce
        float* pt = g.part + ((size_t)sp_ * (g.tiles_m * g.tiles_n) + (size_t)nt_ * g.tiles_m + mt_) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = wm * WM + i * 16 + rgrp + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) pt[ml * BN + wn * WN + j * 16 + col_l] = acc[i][j][r];
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + i * 16 + rgrp + r;
            if (m >= g.Mstore) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 16 + col_l;
                if (n >= g.Nstore) continue;
                int nc = n;
                if (g.cw_cin > 0) { const int tap = n / g.cw_cin, ci = n - tap * g.cw_cin; nc = ci * g.cw_taps + tap; }
                atomicAdd(&g.C[(size_t)m * g.ldc + nc], acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Stem weight gradient (single input channel, 4x4 kernel, 128 output channels):
//   dW[co][t] += sum_m dY[m][co] * patch(m)[t],   db[co] += sum_m dY[m][co]
// is a [128 x 16] GEMM over 1.25 M pixels -- through the generic 128x128 TN tile 7/8 of the B gather and of the MFMAs
// would be padding.  Here a workgroup streams its pixel range in chunks of 32: dY rows (bf16, 256 B each) go to LDS as
// they are, the 32x16 patch values are gathered from the padded fp32 input and stored as bf16 with a 17th column of
// ones (the bias gradient rides along as one more output column), both operands are read back as transposed fragments.
// HBM bound on the single read of dY.  Partials per workgroup, summed by k_stem_wgrad_reduce (no atomics).
// ---------------------------------------------------------------------------------------------
#define STEM_NB 512
__global__ __launch_bounds__(256) void k_stem_wgrad16(const __bf16* __restrict__ dY, const float* __restrict__ xp, float* __restrict__ part,
                                                      int B, int Hin, int s, int Hout, long long M, int chunks_per_block) {
    constexpr int CO = 128, LDA = CO + 8, LDB = 32 + 8;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][32 * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long m_begin = (long long)blockIdx.x * chunks_per_block * 32;
    const long long m_end = min(M, m_begin + (long long)chunks_per_block * 32);
    // constant part of B: column 16 = 1 (bias gradient), 17..31 = 0
    for (int i = tid; i < 2 * 32 * 16; i += 256) {
        const int bufi = i >> 9, r = (i >> 4) & 31, cI = 16 + (i & 15);
        Bs[bufi][r * LDB + cI] = (__bf16)((i & 15) == 0 ? 1.f : 0.f);
    }
    // A loader: thread -> (row = tid>>3, 16 channels at (tid&7)*16)
    const int a_row = tid >> 3, a_c = (tid & 7) * 16;
    // B loader: thread -> (pixel = tid>>3, taps 2*(tid&7), +1): same ky, kx = even
    const int b_px = tid >> 3, b_t = (tid & 7) * 2, b_ky = b_t >> 2, b_kx = b_t & 3;
    // running (b, oy, ox) of this thread's pixel
    long long mp = m_begin + b_px;
    int pb = (int)(mp / ((long long)Hout * Hout));
    int prem = (int)(mp - (long long)pb * Hout * Hout);
    int poy = prem / Hout, pox = prem - poy * Hout;
    uint4 ra0, ra1;
    float rb0, rb1;
    auto load = [&](long long m0) {
        const long long ma = m0 + a_row;
        if (ma < M) {
            const uint4* src = reinterpret_cast<const uint4*>(dY + (size_t)ma * CO + a_c);
            ra0 = src[0]; ra1 = src[1];
        } else {
            ra0 = make_uint4(0, 0, 0, 0); ra1 = ra0;
        }
        const int bb = min(pb, B - 1);                  // tail pixels: any finite value (their dY rows are zero)
        const float* row = xp + ((size_t)bb * Hin + poy * s + b_ky) * Hin + pox * s + b_kx;
        rb0 = row[0]; rb1 = row[1];
        pox += 32;
        while (pox >= Hout) { pox -= Hout; ++poy; }
        while (poy >= Hout) { poy -= Hout; ++pb; }
    };
    auto store = [&](int buf) {
        *reinterpret_cast<uint4*>(&As[buf][a_row * LDA + a_c]) = ra0;
        *reinterpret_cast<uint4*>(&As[buf][a_row * LDA + a_c + 8]) = ra1;
        __bf16 v[2] = {(__bf16)rb0, (__bf16)rb1};
        *reinterpret_cast<unsigned*>(&Bs[buf][b_px * LDB + b_t]) = *reinterpret_cast<unsigned*>(v);
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m_begin < m_end) {
        load(m_begin);
        store(0);
        __syncthreads();
        int buf = 0;
        for (long long m0 = m_begin; m0 < m_end; m0 += 32) {
            const bool more = (m0 + 32) < m_end;
            if (more) load(m0 + 32);
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = lds_tr_frag16(As[buf], LDA, wave * 32 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = lds_tr_frag16(Bs[buf], LDB, j * 16, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (more) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    float* pt = part + (size_t)blockIdx.x * CO * 32;
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) pt[(wave * 32 + i * 16 + rgrp + r) * 32 + j * 16 + col_l] = acc[i][j][r];
}
// dW[co][0..15] += sum_blocks part[blk][co][0..15]; db[co] += sum_blocks part[blk][co][16].  Workgroup = one channel.
__global__ __launch_bounds__(256) void k_stem_wgrad_reduce(const float* __restrict__ part, int nblk, float* __restrict__ dW,
                                                           float* __restrict__ db) {
    __shared__ float red[8][32];
    const int co = blockIdx.x, n = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float t = 0.f;
    for (int blk = grp; blk < nblk; blk += 8) t += part[((size_t)blk * 128 + co) * 32 + n];
    red[grp][n] = t;
    __syncthreads();
    if (grp != 0 || n > 16) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += red[q][n];
    if (n < 16) dW[co * 16 + n] += v;
    else if (db) db[co] += v;
}
int spair_stem_wgrad16_impl(const void* dY, const float* xp, float* dW, float* db, float* part, long long part_cap, int B, int Hin,
                            int s_, int Hout, hipStream_t s) {
    const long long M = (long long)B * Hout * Hout;
    if (part == nullptr || part_cap < (long long)STEM_NB * 128 * 32) return SPAIR_ERR_UNSUPPORTED;
    if ((Hout - 1) * s_ + 4 > Hin) return SPAIR_ERR_SHAPE;
    const long long chunks = (M + 31) / 32;
    const int nblk = (int)min((long long)STEM_NB, chunks);
    const int cpb = (int)((chunks + nblk - 1) / nblk);
    hipLaunchKernelGGL(k_stem_wgrad16, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dY), xp, part, B, Hin, s_, Hout, M, cpb);
    SPAIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3(128), dim3(256), 0, s, part, nblk, dW, db);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// A bf16 [R][lda]; B: bf16 (b_bf16) plain rows / conv gather, or fp32 conv gather / plain rows
int spair_gemm_tn16_impl(GemmTN g, bool conv, bool b_bf16, hipStream_t s) {
    if (g.ngroup > 1) { g.M = 128; g.N = 128; g.Mstore = 128; g.Nstore = 128; g.lda = g.tile[0].lda; g.ldb = g.tile[0].ldb; }
    if (g.M <= 0 || g.N <= 0 || g.R <= 0) return SPAIR_ERR_SHAPE;
    if (g.Mstore <= 0) g.Mstore = g.M;
    if (g.Nstore <= 0) g.Nstore = g.N;
    if ((g.lda & 7) || (g.M & 7)) return SPAIR_ERR_ALIGN;
    if (b_bf16 && ((g.N & 7) || (!conv && (g.ldb & 7)) || (conv && (g.conv.Cin & 7)))) return SPAIR_ERR_ALIGN;
    if (!b_bf16 && ((g.N & 3) || (!conv && (g.ldb & 3)))) return SPAIR_ERR_ALIGN;
    if ((long long)g.R * g.lda >= (1ll << 31) || (!conv && (long long)g.R * g.ldb >= (1ll << 31))) return SPAIR_ERR_UNSUPPORTED;   // 32-bit offsets
    constexpr int BM = 128, BN = 128;
    if (g.ngroup > 1) {   // grouped single-tile problems: only through the partial-tile path, plain rows
        if (g.ngroup > SPAIR_TN_MAX_TILES || conv || !g.part) return SPAIR_ERR_UNSUPPORTED;
        for (int q = 0; q < g.ngroup; ++q) {
            const GemmTN::Tile& t = g.tile[q];
            if (!t.A || !t.B || !t.C || t.M <= 0 || t.N <= 0 || t.M > BM || t.N > BN || (t.M & 7) || (t.lda & 7)) return SPAIR_ERR_SHAPE;
            if (b_bf16 ? ((t.N & 7) || (t.ldb & 7)) : ((t.N & 3) || (t.ldb & 3))) return SPAIR_ERR_ALIGN;
            if ((long long)g.R * t.lda >= (1ll << 31) || (long long)g.R * t.ldb >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
            if (t.Mstore <= 0 || t.m_skip < 0 || t.m_skip + t.Mstore > t.M || t.Nstore <= 0 || t.n_skip < 0 || t.n_skip + t.Nstore > t.N) return SPAIR_ERR_SHAPE;
        }
        g.M = BM; g.N = BN; g.Mstore = BM; g.Nstore = BN; g.lda = g.tile[0].lda; g.ldb = g.tile[0].ldb;
    }
    if (b_bf16) {      // DMA-staged loader / consumer kernel (tn_ring.hip) wherever it applies
        const int rc = spair_gemm_tn_ring(g, conv, s);
        if (rc != SPAIR_ERR_UNSUPPORTED) return rc;
    }
    const int tiles = ceil_div(g.M, BM) * ceil_div(g.N, BN) * (g.ngroup > 1 ? g.ngroup : 1);
    // one full round of resident blocks: 256 CUs x 3 blocks (150 VGPRs, 35 KB LDS); a 4/3-round grid wastes a third of the time
    int nsplit = max(1, min(ceil_div(g.R, 256), 768 / tiles));
    if (nsplit >= 8) nsplit = nsplit / 8 * 8;
    int rps = round_up(ceil_div(g.R, nsplit), 32);
    if (ceil_div(g.R, rps) != nsplit) nsplit = ceil_div(g.R, rps);
    g.rows_per_split = rps; g.nsplit = nsplit; g.tiles_m = ceil_div(g.M, BM); g.tiles_n = g.ngroup > 1 ? g.ngroup : ceil_div(g.N, BN);
    dim3 grid(g.tiles_m * g.tiles_n * nsplit);
    if (g.part && (long long)grid.x * (BM * BN + BM) > g.part_cap) { if (g.ngroup > 1) return SPAIR_ERR_UNSUPPORTED; g.part = nullptr; }   // scratch too small: atomics
    g.colpart = g.part ? g.part + (size_t)grid.x * BM * BN : nullptr;     // per-block column sums (bias gradients) behind the partial tiles
    if (conv) {
        if (b_bf16) hipLaunchKernelGGL((gemm_tn16_kernel<true, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_tn16_kernel<true, false>), grid, dim3(256), 0, s, g);
    } else {
        if (b_bf16) hipLaunchKernelGGL((gemm_tn16_kernel<false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_tn16_kernel<false, false>), grid, dim3(256), 0, s, g);
    }
    SPAIR_CHECK_LAUNCH();
    if (g.part) return spair_tn_reduce(g, BM, BN, s);
    return SPAIR_OK;
}

// fp32 -> bf16 row-major copy with leading dimensions (d feat for the backbone's backward pass)
__global__ __launch_bounds__(256) void k_to_bf16(const float* __restrict__ src, int lds_, __bf16* __restrict__ dst, int ldd, long long rows, int cols) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const long long r = idx / cols;
    const int c = (int)(idx - r * cols);
    dst[r * ldd + c] = (__bf16)src[r * lds_ + c];
}
int spair_to_bf16(const float* src, int lds_, void* dst, int ldd, long long rows, int cols, hipStream_t s) {
    const long long n = rows * cols;
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, lds_, reinterpret_cast<__bf16*>(dst), ldd, rows, cols);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// ---- unit-level C ABI (bf16 tensors are passed as opaque device pointers) --------------------------------------------------
extern "C" int spair_gemm_nt16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                               const void* relu_mask, int ldmask, int mask_bf16, int relu, int c_bf16, const int* conv13, const int* cmap8,
                               void* stream) {
    GemmNT g{};
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.B = B; g.ldb = ldb; g.C = reinterpret_cast<float*>(C); g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.mask = reinterpret_cast<const float*>(relu_mask); g.ldmask = ldmask; g.mask_bf16 = mask_bf16;
    g.relu = relu; g.c_bf16 = c_bf16;
    if (conv13) {
        const int* p = conv13;
        g.conv.Hin = p[0]; g.conv.Win = p[1]; g.conv.Cin = p[2]; g.conv.Hout = p[3]; g.conv.Wout = p[4]; g.conv.kh = p[5]; g.conv.kw = p[6];
        g.conv.sy = p[7]; g.conv.sx = p[8]; g.conv.dky = p[9]; g.conv.dkx = p[10]; g.conv.oy = p[11]; g.conv.ox = p[12];
    }
    if (cmap8) {
        g.use_cmap = 1;
        g.cmap.Hout = cmap8[0]; g.cmap.Wout = cmap8[1]; g.cmap.Hc = cmap8[2]; g.cmap.Wc = cmap8[3];
        g.cmap.osy = cmap8[4]; g.cmap.osx = cmap8[5]; g.cmap.ooy = cmap8[6]; g.cmap.oox = cmap8[7];
    }
    return spair_gemm_nt16_impl(g, conv13 != nullptr, (hipStream_t)stream);
}

extern "C" int spair_gemm_tn16(const void* A, int lda, const void* B, int ldb, int b_bf16, float* C, int ldc, int M, int N, int R,
                               const int* conv13, int cw_cin, int cw_taps, float* colsum_out, float* scratch, long long scratch_floats,
                               void* stream) {
    GemmTN g{};
    g.part = scratch; g.part_cap = scratch_floats;
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.B = reinterpret_cast<const float*>(B); g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = round_up(M, 8); g.N = round_up(N, b_bf16 ? 8 : 4); g.Mstore = M; g.Nstore = N; g.R = R; g.cw_cin = cw_cin; g.cw_taps = cw_taps;
    g.colsum_out = colsum_out;
    if (conv13) {
        const int* p = conv13;
        g.conv.Hin = p[0]; g.conv.Win = p[1]; g.conv.Cin = p[2]; g.conv.Hout = p[3]; g.conv.Wout = p[4]; g.conv.kh = p[5]; g.conv.kw = p[6];
        g.conv.sy = p[7]; g.conv.sx = p[8]; g.conv.dky = p[9]; g.conv.dkx = p[10]; g.conv.oy = p[11]; g.conv.ox = p[12];
    }
    return spair_gemm_tn16_impl(g, conv13 != nullptr, b_bf16 != 0, (hipStream_t)stream);
}

extern "C" int spair_stem_wgrad16(const void* dY, const float* xpad, float* dW, float* db, float* scratch, long long scratch_floats, int B,
                                  int Hin, int stride, int Hout, void* stream) {
    return spair_stem_wgrad16_impl(dY, xpad, dW, db, scratch, scratch_floats, B, Hin, stride, Hout, (hipStream_t)stream);
}
extern "C" int spair_cast_bf16(const float* src, int ld_src, void* dst, int ld_dst, long long rows, int cols, void* stream) {
    return spair_to_bf16(src, ld_src, dst, ld_dst, rows, cols, (hipStream_t)stream);
}
