// Weight-gradient GEMM  C[m][n] (+)= sum_r A[r][m] . B[r][n]  of the bf16 step (modules.py's Linear / Conv2d weight gradients: A = the layer's
// output gradient, B = its input rows, or the implicit im2col gather of them), as a split-K kernel whose operands go HBM/L2 -> LDS by DMA only.
//
// gemm16.hip's gemm_tn16_kernel stages 32 rows per step through registers (global load -> VGPR -> ds_write, one barrier per 16 MFMAs per wave)
// and sat at 12-26 % of the matrix-core peak on every weight gradient of the step.  Here the roles are split as in conv_s2.hip:
//   * 4 LOADER waves issue every LDS-DMA (global -> LDS, 1 KB per instruction) behind counted waits: a stage is 64 K rows of A (128 columns)
//     and of B (BN = 128 or 256 columns), a ring of 3 stages (96 / 144 KB of LDS, one workgroup per CU);
//   * 4 COMPUTING waves (2 x 2, 64 x BN/2 outputs each) read transposed fragments (ds_read_b64_tr_b16) and run 32 / 64 MFMAs per stage
//     behind ONE workgroup barrier.
// Stage layout: [64-column group][64 rows][128 B]; the eight 16-byte chunks of a row segment are stored pair-swizzled, pair ^ key(row) with
// key(row) = (row & 3) ^ ((row >> 3) & 1), applied on the SOURCE side of the DMA (its LDS side is always wave base + lane * 16).  A transposing
// read (ds_read_b64_tr_b16) is served in two groups of 32 lanes, each 8 rows {r, r+1, r+2, r+3, r+8, ..., r+11} x 32 B against 64 banks of 4 B:
// with the key the eight 32-byte pieces fall on eight different bank groups (2 LDS cycles per read, the minimum).  With pair ^ (row & 3) alone
// rows r and r + 8 shared their banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 (profiles/r03_b_sq_counters.txt).  Rows past the split's end and columns past the operand's extent are requested out
// of range of the buffer descriptor: the DMA writes zeros.
// Output: the fp32 partial tile of this split (+ the column sums of A = the bias gradient, taken with one extra MFMA against a fragment of
// ones), summed over the splits in a fixed order by k_tn_reduce (gemm.hip) -- no atomics, gradients repeat bit for bit.
#include "common.h"
#include "gemm.h"

namespace {

constexpr int TR_BM = 128, TR_BK = 64;
constexpr int TR_GRP_B = TR_BK * 128;            // one 64-column group of a stage: 64 rows x 128 B

typedef short tr_v4s16 __attribute__((ext_vector_type(4)));
template <int W>
__device__ __forceinline__ void tr_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory"); }

// rows R, R + 4 (8 bytes each) of a swizzled group -> one 32 (K) x 16 (columns) operand fragment
__device__ __forceinline__ bf16x8 tr_frag(const char* p) {
    const tr_v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s16*)(p));
    const tr_v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s16*)(p + 4 * 128));
    union { struct { tr_v4s16 lo, hi; } s; bf16x8 v; } u;
    u.s.lo = lo; u.s.hi = hi;
    return u.v;
}

template <int BN, bool BCONV, int NST>
__global__ __launch_bounds__(512) void k_tn_ring(GemmTN g) {
    constexpr int NGB = BN / 64;                      // 64-column groups of B
    constexpr int STAGE_B = (2 + NGB) * TR_GRP_B;     // A: groups 0, 1; B: groups 2 ..
    constexpr int PER = 2 * (2 + NGB);                // DMA instructions per loader wave and stage
    constexpr int TN = BN / 32;                       // 16-column fragments per computing wave
    extern __shared__ __attribute__((aligned(16))) char tr_sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mt_, nt_, sp_;
    {
        const int id = blockIdx.x, ntm = g.tiles_m, ntn = g.tiles_n;
        if ((g.nsplit & 7) == 0) {   // XCD-aware: all tiles of one row split on one XCD
            const int xcd = id & 7, j = id >> 3;
            mt_ = j % ntm; nt_ = (j / ntm) % ntn; sp_ = (j / (ntm * ntn)) * 8 + xcd;
        } else {
            mt_ = id % ntm; nt_ = (id / ntm) % ntn; sp_ = id / (ntm * ntn);
        }
    }
    const bool grouped = g.ngroup > 1;
    const GemmTN::Tile& gt = g.tile[grouped ? nt_ : 0];
    const void* gA = grouped ? gt.A : reinterpret_cast<const void*>(g.A);
    const void* gB = grouped ? gt.B : reinterpret_cast<const void*>(g.B);
    const int g_lda = grouped ? gt.lda : g.lda, g_ldb = grouped ? gt.ldb : g.ldb;
    const int g_M = grouped ? gt.M : g.M, g_N = grouped ? gt.N : g.N;
    float* g_colsum = grouped ? gt.colsum : g.colsum_out;
    const int m0 = mt_ * TR_BM, n0 = grouped ? 0 : nt_ * BN;
    const int r_begin = sp_ * g.rows_per_split;
    const int r_end = min(g.R, r_begin + g.rows_per_split);
    const int nst = r_begin < r_end ? (r_end - r_begin + TR_BK - 1) / TR_BK : 0;
    const int tiles = g.tiles_m * g.tiles_n, tile = nt_ * g.tiles_m + mt_;

    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves: rows 16 lw + 8 i + (lane >> 3), i = 0, 1, of every group
        const int lw = wave - 4;
        const int rsub = lane >> 3, pos = lane & 7;
        // the source chunk this lane fetches in the pieces of its row i: its rows are 16 lw + 8 i + rsub, so key = (rsub & 3) ^ i
        const __amdgpu_buffer_rsrc_t rsA = buf_rsrc(gA), rsB = buf_rsrc(gB);
        unsigned a_col[2][2], b_col[2][NGB];
        bool a_ok[2][2], b_ok[2][NGB];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = pos ^ ((((rsub & 3) ^ i) & 3) << 1);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) { const int m = m0 + gi * 64 + c * 8; a_ok[i][gi] = m < g_M; a_col[i][gi] = (unsigned)m; }
#pragma unroll
            for (int gi = 0; gi < NGB; ++gi) {
                const int n = n0 + gi * 64 + c * 8;
                b_ok[i][gi] = n < g_N;
                if (BCONV) {      // n = tap * Cin + ci -> offset of that tap's pixel and channel from the output pixel's first input element
                    const int nn = min(n, g_N - 8), tap = nn / g.conv.Cin, ci = nn - tap * g.conv.Cin, ky = tap / g.conv.kw, kx = tap - ky * g.conv.kw;
                    b_col[i][gi] = (unsigned)((ky * g.conv.dky * g.conv.Win + kx * g.conv.dkx) * g.conv.Cin + ci);
                } else {
                    b_col[i][gi] = (unsigned)n;
                }
            }
        }
        const int HW = BCONV ? g.conv.Hout * g.conv.Wout : 1;
        auto glds = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte_off, char* dst) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)byte_off, 0, 0, 0);
        };
        auto issue = [&](int st) {
            char* sb = tr_sm + (st % NST) * STAGE_B;
            const bool live = st < nst;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = 16 * lw + 8 * i, row = r_begin + st * TR_BK + rl + rsub;
                const bool rok = live & (row < r_end);       // (bitwise on purpose: no short-circuit control flow around the DMAs)
                const unsigned ra = (unsigned)row * (unsigned)g_lda;
                unsigned rb;
                if (BCONV) {      // (computed for masked rows too: a conditional here makes hipcc split the DMAs below by lane mask, and the
                                  //  counted waits need every wave to issue exactly PER of them per stage)
                    const unsigned rr = (unsigned)row;
                    const unsigned b = rr / (unsigned)HW, rem = rr - b * (unsigned)HW, y = rem / (unsigned)g.conv.Wout, x = rem - y * (unsigned)g.conv.Wout;
                    rb = (unsigned)(((b * g.conv.Hin + y * g.conv.sy + g.conv.oy) * g.conv.Win + x * g.conv.sx + g.conv.ox) * g.conv.Cin);
                } else {
                    rb = (unsigned)row * (unsigned)g_ldb;
                }
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) glds(rsA, (rok & a_ok[i][gi]) ? (ra + a_col[i][gi]) * 2u : BUF_OOB, sb + gi * TR_GRP_B + rl * 128);
#pragma unroll
                for (int gi = 0; gi < NGB; ++gi)
                    glds(rsB, (rok & b_ok[i][gi]) ? (rb + b_col[i][gi]) * 2u : BUF_OOB, sb + (2 + gi) * TR_GRP_B + rl * 128);
            }
        };
#pragma unroll
        for (int st = 0; st < NST - 1; ++st) issue(st);
        tr_wait<PER * (NST - 2)>();                  // stage 0 has landed
#pragma unroll 1
        for (int kt = 0; kt < nst; ++kt) {
            __builtin_amdgcn_s_barrier();            // barrier(kt): stage kt is complete for everyone; the computing waves are done with stage kt - 1
            asm volatile("" ::: "memory");
            issue(kt + NST - 1);                     // into the slot of stage kt - 1 (zeros past the last stage: keeps the counts uniform)
            tr_wait<PER * (NST - 2)>();              // stage kt + 1 has landed
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing DMAs target this workgroup's LDS
        return;
    }

    // ---------------------------------------------------------------- computing waves
    const int wm = wave >> 1, wn = wave & 1;
    const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int rowoff = (8 * g4 + q) * 128 + (p << 3);
    int poff[4];                                     // pair j of a group sits at pair position j ^ key(row); the lane's rows 8 g4 + q (+ 4, + 32): key = q ^ (g4 & 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) poff[j] = rowoff + ((j ^ q ^ (g4 & 1)) << 5);
    const int a_base = wm * TR_GRP_B;                              // the wave's 64 rows of C = one group of A
    const int b_base = (2 + wn * (NGB / 2)) * TR_GRP_B;            // its BN / 2 columns = NGB / 2 groups of B
    const bool do_colsum = g_colsum != nullptr && (grouped || nt_ == 0) && wn == 0;
    f32x4 acc[4][TN], cacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

#pragma unroll 1
    for (int kt = 0; kt < nst; ++kt) {
        __builtin_amdgcn_s_barrier();                // barrier(kt): stage kt has landed
        asm volatile("" ::: "memory");
        const char* sb = tr_sm + (kt % NST) * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[TN];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = tr_frag(sb + a_base + ks * 32 * 128 + poff[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = tr_frag(sb + b_base + (j >> 2) * TR_GRP_B + ks * 32 * 128 + poff[j & 3]);
#pragma unroll
            for (int j = 0; j < TN; ++j)             // B fragment outermost: the first MFMAs need the 4 A fragments and ONE B fragment only
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, cacc[i], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    // ---- this split's partial tile (C layout: lane holds column j * 16 + (lane & 15), rows i * 16 + (lane >> 4) * 4 + r) and column sums
    const int col_l = lane & 15, rgrp = (lane >> 4) * 4;
    float* pt = g.part + ((size_t)sp_ * tiles + tile) * (size_t)(TR_BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = wm * 64 + i * 16 + rgrp + r;
#pragma unroll
            for (int j = 0; j < TN; ++j) pt[ml * BN + wn * (BN / 2) + j * 16 + col_l] = acc[i][j][r];
        }
    if (g_colsum != nullptr && (grouped || nt_ == 0) && wn == 0 && col_l == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) g.colpart[((size_t)sp_ * tiles + tile) * TR_BM + wm * 64 + i * 16 + rgrp + r] = cacc[i][r];
    }
}

template <int BN, bool BCONV, int NST>
int tr_launch(const GemmTN& g, dim3 grid, hipStream_t s) {
    constexpr int lds = NST * (2 + BN / 64) * TR_GRP_B;
    static std::atomic<unsigned long long> attr_done{0};
    if (spair_dyn_lds_once(reinterpret_cast<const void*>(&k_tn_ring<BN, BCONV, NST>), lds, attr_done) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    hipLaunchKernelGGL((k_tn_ring<BN, BCONV, NST>), grid, dim3(512), lds, s, g);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

}  // namespace

// Called by spair_gemm_tn16_impl (gemm16.hip) with a validated problem (bf16 B, load extents multiples of 8, 32-bit element offsets); returns
// SPAIR_ERR_UNSUPPORTED when the ring kernel does not apply and the caller keeps gemm_tn16_kernel.
int spair_gemm_tn_ring(GemmTN g, bool conv, hipStream_t s) {
    if (!g.part) return SPAIR_ERR_UNSUPPORTED;
    const bool grouped = g.ngroup > 1;
    if (conv) {
        if (grouped || (g.conv.Cin & 63) || g.conv.kw <= 0) return SPAIR_ERR_UNSUPPORTED;
        const long long rows_in = (long long)ceil_div(g.R, g.conv.Hout * g.conv.Wout) * g.conv.Hin * g.conv.Win;
        if (rows_in * g.conv.Cin >= (1ll << 31)) return SPAIR_ERR_UNSUPPORTED;
    }
    const int BN = (!grouped && g.N > 128) ? 256 : 128;
    const int tiles_m = ceil_div(g.M, TR_BM), tiles_n = grouped ? g.ngroup : ceil_div(g.N, BN);
    const int tiles = tiles_m * tiles_n;
    const int stages = ceil_div(g.R, TR_BK);
    if (stages < 4) return SPAIR_ERR_UNSUPPORTED;
    // one workgroup per CU (96 / 144 KB of LDS): one round of 256, every split at least 2 stages
    int nsplit = std::max(1, std::min(stages / 2, 256 / std::max(1, tiles)));
    if (nsplit >= 8) nsplit = nsplit / 8 * 8;       // a multiple of 8: the kernel then keeps every tile of a row split on one XCD (shared operand slabs hit its L2)
    int sps = ceil_div(stages, nsplit);
    nsplit = ceil_div(stages, sps);
    g.rows_per_split = sps * TR_BK; g.nsplit = nsplit; g.tiles_m = tiles_m; g.tiles_n = tiles_n;
    const dim3 grid((unsigned)(tiles * nsplit));
    if ((long long)grid.x * (TR_BM * BN + TR_BM) > g.part_cap) return SPAIR_ERR_UNSUPPORTED;
    g.colpart = g.part + (size_t)grid.x * TR_BM * BN;
    int rc;
    // ring of 3 stages (measured at BN = 128, where 4 and 5 fit: no change -- the stage time is set by the DMA issue rate, not its latency)
    if (BN == 256) rc = conv ? tr_launch<256, true, 3>(g, grid, s) : tr_launch<256, false, 3>(g, grid, s);
    else rc = conv ? tr_launch<128, true, 3>(g, grid, s) : tr_launch<128, false, 3>(g, grid, s);
    if (rc != SPAIR_OK) return rc;
    return spair_tn_reduce(g, TR_BM, BN, s);
}
