// The object decoder's DATA-GRADIENT chain of the bf16 step (backward of models.py:474-492's three Linear layers; the sprite sigmoid's
// derivative is already in d-logits) as ONE kernel:
//     dH2 = (dL . W2) * [H2 > 0]      dH1 = (dH2 . W1) * [H1 > 0]      d z_attr = dH1 . W0
// Three gemm_nt16 launches took 0.098 + 0.024 + 0.022 ms (the first at 0.21 of the matrix-core peak); the first product is 52.6 of the 58.6
// GFLOP and is bound by what a CU can pull into LDS, so it is laid out like the weight-gradient kernel (tn_ring.hip):
//   * a stage = 64 k of W2^T (256 rows = H2 channels, 32 KB, L2-resident), ring of 3 in LDS; 4 LOADER waves issue every LDS-DMA behind
//     counted waits, 4 COMPUTING waves own 32 complete rows each and fetch their d-logit fragments straight from HBM into registers;
//   * the products are computed TRANSPOSED (weights = A operand, rows = B operand) with the weight rows of a tile pair read in the order
//     n(T, m) = 32 (T >> 1) + 8 (m >> 2) + 4 (T & 1) + (m & 3): a lane ends up with 8 consecutive channels of one row, so the relu gate and
//     the store of dH2 work on registers, 16 bytes at a time, and the gated bf16x8 IS the B fragment of the next layer's K step;
//   * layers 1 and 0 continue activation-stationary on those registers: their transposed weights arrive as three more ring stages
//     (W1^T: 2 x [2 k-blocks][128 rows][64 k]; W0^T: [2 k-blocks][64 rows][64 k]) while the first epilogue runs.
// LDS rows are 128 B with the 16-byte chunks XOR-swizzled on the DMA's source side: weight rows by (n & 3) | (((n >> 3) & 1) << 2) (8 keys
// over the 16 permuted rows of a fragment): conflict-free for ds_read_b128's 16-lane groups.
#include "common.h"
#include "dec_fused.h"

namespace {

constexpr int DB_ROWS = 128;                     // rows per workgroup, 32 per computing wave
constexpr int DB_H1 = 128, DB_H2 = 256;
constexpr int DB_W_B = 256 * 128, DB_STAGE_B = DB_W_B;      // 32 KB: the weight tile only (the d-logit rows go straight to registers)
constexpr int DB_NST = 3;
constexpr int DB_PER = 8;                        // DMA instructions per loader wave and stage

struct DecBwdArgs {
    const u16* dL; int ld_s;                     // d-logits [N][ld_s] bf16, K1 = n_out columns
    const u16* W2t; int ld2;                     // W2^T [256][ld2] bf16 (k = logit)
    const u16* W1t;                              // W1^T [128][256]
    const u16* W0t; int A;                       // W0^T [A][128], A <= 64
    const u16* H2; const u16* H1;                // stored activations (relu gates) [N][256], [N][128] bf16
    u16* dH2; u16* dH1;                          // out, same shapes
    float* dza; int ld_dza;                      // out: d z_attr fp32 [N][ld_dza], A columns
    int N, K1;
};

__device__ __forceinline__ int db_wkey(int n) { return (n & 3) | (((n >> 3) & 1) << 2); }
template <int W>
__device__ __forceinline__ void db_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory"); }

// 8 bf16 of v kept where the matching bf16 of g is > 0 (positive and not zero)
__device__ __forceinline__ uint4 db_gate8(uint4 v, uint4 g) {
    unsigned* pv = reinterpret_cast<unsigned*>(&v);
    const unsigned* pg = reinterpret_cast<const unsigned*>(&g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = pg[i] & 0xffffu, hi = pg[i] >> 16;
        const unsigned keep = (((lo & 0x7fffu) != 0u && !(lo & 0x8000u)) ? 0xffffu : 0u) | (((hi & 0x7fffu) != 0u && !(hi & 0x8000u)) ? 0xffff0000u : 0u);
        pv[i] &= keep;
    }
    return v;
}
__device__ __forceinline__ uint4 db_pack8(const f32x4& a, const f32x4& b) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (__bf16)a[e]; v[4 + e] = (__bf16)b[e]; }
    uint4 o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}
__device__ __forceinline__ bf16x8 db_frag(const uint4& v) {
    bf16x8 o;
    __builtin_memcpy(&o, &v, 16);
    return o;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dec_bwd(DecBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char db_sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * DB_ROWS;
    const int nk1 = (a.K1 + 63) / 64;            // stages of the first layer
    const int nst = nk1 + 3;                     // + 2 (W1^T) + 1 (W0^T)

    if (wave >= 4) {
        // ------------------------------------------------------------ loader waves: pieces of 8 rows x 128 B
        const int lw = wave - 4;
        const int rsub = lane >> 3, pos = lane & 7;
        const __amdgpu_buffer_rsrc_t r2 = buf_rsrc(a.W2t), r1 = buf_rsrc(a.W1t), r0 = buf_rsrc(a.W0t);
        auto glds = [&](__amdgpu_buffer_rsrc_t rs, unsigned byte_off, char* dst) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)byte_off, 0, 0, 0);
        };
        // per piece (fixed LDS row n, source chunk c): byte offsets of the three weight matrices; selected by the (wave-uniform) stage and
        // masked with bitwise conditions only -- a short-circuit or a branch here makes hipcc split one DMA into lane-masked ones, and
        // the counted waits need exactly DB_PER of them per stage and wave
        const int K1 = a.K1;
        unsigned w2off[8], w1off[8], w0off[8], c8w[8];
        bool w0ok[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = (lw * 8 + e) * 8 + rsub;
            const int c = pos ^ db_wkey(n);
            c8w[e] = (unsigned)(c * 8);
            w2off[e] = ((unsigned)n * (unsigned)a.ld2 + (unsigned)(c * 8)) * 2u;                                   // + stage * 128
            w1off[e] = ((unsigned)(n & 127) * DB_H2 + (unsigned)((n >> 7) * 64 + c * 8)) * 2u;                    // + (stage - nk1) * 256
            w0off[e] = ((unsigned)(n & 63) * DB_H1 + (unsigned)(((n >> 6) & 1) * 64 + c * 8)) * 2u;
            w0ok[e] = (n < 128) & ((n & 63) < a.A);
        }
        auto issue = [&](int st) __attribute__((always_inline)) {
            char* sb = db_sm + (st % DB_NST) * DB_STAGE_B;
            const unsigned kst = (unsigned)st * 64u;
            // three wave-uniform paths, each issuing exactly 8 weight pieces (a select between the buffer descriptors would be a waterfall loop)
            if (st < nk1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) glds(r2, ((kst + c8w[e]) < (unsigned)K1) ? w2off[e] + kst * 2u : BUF_OOB, sb + (lw * 8 + e) * 1024);
            } else if (st < nk1 + 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) glds(r1, w1off[e] + (unsigned)(st - nk1) * 256u, sb + (lw * 8 + e) * 1024);
            } else {
                const bool l0 = st == nk1 + 2;
#pragma unroll
                for (int e = 0; e < 8; ++e) glds(r0, (l0 & w0ok[e]) ? w0off[e] : BUF_OOB, sb + (lw * 8 + e) * 1024);
            }
        };
        issue(0);
        issue(1);
        db_wait<DB_PER>();                       // stage 0 has landed
#pragma unroll 1
        for (int kt = 0; kt < nst; ++kt) {
            __builtin_amdgcn_s_barrier();        // barrier(kt): stage kt is complete; the computing waves are done with stage kt - 1
            asm volatile("" ::: "memory");
            issue(kt + 2);                       // (zeros past the last stage: keeps the counts uniform)
            db_wait<DB_PER>();                   // stage kt + 1 has landed
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---------------------------------------------------------------- computing waves: rows m0 + 32 wave + 16 j + r16, j = 0, 1
    const int q = lane >> 4, r16 = lane & 15;
    const int wkey = (r16 & 3) | (((r16 >> 2) & 1) << 2);
    const int wrow = (r16 >> 2) * 8 + (r16 & 3);                 // + 32 (T >> 1) + 4 (T & 1): LDS row of fragment tile T for this lane
    bool rok[2];
    unsigned grow[2], xg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = wave * 32 + j * 16 + r16;
        rok[j] = (m0 + m) < a.N;
        grow[j] = (unsigned)(m0 + m);
        xg[j] = ((unsigned)(m0 + m) * (unsigned)a.ld_s + (unsigned)(q * 8)) * 2u;      // + k-step * 64 bytes
    }
    // The d-logit rows never touch LDS: a wave owns its 32 rows, so the B fragment of (row tile j, k-step) -- row = lane & 15, 8 k at
    // 8 (lane >> 4) -- is one 16-byte buffer load per lane straight from the row-major tensor, requested a stage ahead.  That takes a third
    // of the LDS-DMA instructions out of the loaders' stream, whose issue rate (12 pieces of 1 KiB per wave and us) is what bounds a stage.
    const __amdgpu_buffer_rsrc_t rx = buf_rsrc(a.dL);
    const int K1 = a.K1;
    auto xload = [&](int kt, uint4 (&x)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = kt * 64 + ks * 32 + q * 8;
                x[ks * 2 + j] = buf_load16(rx, (rok[j] & (k < K1) & (kt < nk1)) ? xg[j] + (unsigned)(kt * 128 + ks * 64) : BUF_OOB);
            }
    };
    auto wfrag = [&](const char* sb, int row_base, int T, int c) -> bf16x8 {
        const int n = row_base + wrow + (T >> 1) * 32 + (T & 1) * 4;
        return *reinterpret_cast<const bf16x8*>(sb + n * 128 + ((c ^ wkey) << 4));
    };

    const __amdgpu_buffer_rsrc_t rH2 = buf_rsrc(a.H2), rdH2 = buf_rsrc(a.dH2), rH1 = buf_rsrc(a.H1), rdH1 = buf_rsrc(a.dH1);
    // ---- layer 2 (decoder.out): dH2^T[256 ch][32 rows] over K1
    f32x4 acc[16][2];
#pragma unroll
    for (int T = 0; T < 16; ++T)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[T][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K loop.  Weight fragments: the 16 of a k-step are read in four groups of 4, each group one ahead of its 8 MFMAs (left to itself hipcc
    // keeps two fragments in flight and waits for each pair: an LDS round trip per 4 MFMAs).  d-logit fragments: requested TWO stages ahead
    // (48 registers in flight): they come from HBM, and one stage (~1 us) does not cover that round trip under load.
    bf16x8 wA[4], wB[4];
    auto read_grp = [&](const char* sb, int c, int g, bf16x8 (&w)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = wfrag(sb, 0, g * 4 + t, c);
    };
    auto mma_grp = [&](int g, const bf16x8 (&w)[4], const bf16x8 (&x)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[g * 4 + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t], x[j], acc[g * 4 + t][j], 0, 0, 0);
    };
    // (three register sets in static rotation: moving a set into another would wait for its pending loads -- the whole prefetch)
    uint4 xb0[4], xb1[4], xb2[4];
    auto stage = [&](int kt, uint4 (&xcur)[4], uint4 (&xfill)[4]) __attribute__((always_inline)) {
        xload(kt + 2, xfill);                    // (out of range behind the last stage: zeros, never used)
        __builtin_amdgcn_s_barrier();            // barrier(kt)
        asm volatile("" ::: "memory");
        const char* sb = db_sm + (kt % DB_NST) * DB_STAGE_B;
        const bf16x8 x0[2] = {db_frag(xcur[0]), db_frag(xcur[1])}, x1[2] = {db_frag(xcur[2]), db_frag(xcur[3])};
        read_grp(sb, q, 0, wA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {            // group g of 8: k-step g >> 2, fragments 4 (g & 3) ..; the next group's reads go first
            const int gn = g + 1;
            if (gn < 8) { if (gn & 1) read_grp(sb, (gn >> 2) * 4 + q, gn & 3, wB); else read_grp(sb, (gn >> 2) * 4 + q, gn & 3, wA); }
            __builtin_amdgcn_sched_barrier(0);
            if (g & 1) mma_grp(g & 3, wB, (g >> 2) ? x1 : x0); else mma_grp(g & 3, wA, (g >> 2) ? x1 : x0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): (already true) every read of the stage is in registers before the next barrier
    };
    xload(0, xb0);
    xload(1, xb1);
    int kt = 0;
#pragma unroll 1
    for (; kt + 3 <= nk1; kt += 3) {
        stage(kt, xb0, xb2);
        stage(kt + 1, xb1, xb0);
        stage(kt + 2, xb2, xb1);
    }
    if (kt < nk1) stage(kt, xb0, xb2);
    if (kt + 1 < nk1) stage(kt + 1, xb1, xb0);
    // ---- epilogue 2: relu gate of H2, store dH2, keep it as the B fragments of layer 1 (K step p = channels 32 p .. 32 p + 31)
    uint4 g2[2][8];                              // relu gates: all 16 pieces requested before the first is used
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) g2[j][p] = buf_load16(rH2, rok[j] ? (grow[j] * DB_H2 + (unsigned)(p * 32 + q * 8)) * 2u : BUF_OOB);
    __builtin_amdgcn_sched_barrier(0);
    uint4 g1[2][4];                              // H1's gates: requested here, used behind layer 1's 128 MFMAs
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) g1[j][p] = buf_load16(rH1, rok[j] ? (grow[j] * DB_H1 + (unsigned)(p * 32 + q * 8)) * 2u : BUF_OOB);
    uint4 h2f[2][8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = rok[j] ? (grow[j] * DB_H2 + (unsigned)(p * 32 + q * 8)) * 2u : BUF_OOB;
            const uint4 v = db_gate8(db_pack8(acc[2 * p][j], acc[2 * p + 1][j]), g2[j][p]);
            h2f[j][p] = v;
            buf_store16(rdH2, off, v);
        }
    // ---- layer 1: dH1^T[128 ch][32 rows], K = 256 = ring stages nk1, nk1 + 1 (two k blocks of 64 each)
    f32x4 acc1[8][2];
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[T][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        __builtin_amdgcn_s_barrier();            // barrier(nk1 + s2)
        asm volatile("" ::: "memory");
        const char* sb = db_sm + ((nk1 + s2) % DB_NST) * DB_STAGE_B;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {         // K step p = 4 s2 + kk: k block kk >> 1 of this stage (rows + 128), chunk (kk & 1) * 4 + q
            const int p = 4 * s2 + kk;
#pragma unroll
            for (int T = 0; T < 8; ++T) {
                const bf16x8 wf = wfrag(sb, (kk >> 1) * 128, T, (kk & 1) * 4 + q);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, db_frag(h2f[j][p]), acc1[T][j], 0, 0, 0);
            }
        }
    }
    uint4 h1f[2][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = rok[j] ? (grow[j] * DB_H1 + (unsigned)(p * 32 + q * 8)) * 2u : BUF_OOB;
            const uint4 v = db_gate8(db_pack8(acc1[2 * p][j], acc1[2 * p + 1][j]), g1[j][p]);
            h1f[j][p] = v;
            buf_store16(rdH1, off, v);
        }
    // ---- layer 0: d z_attr^T[A <= 64][32 rows], K = 128 = ring stage nk1 + 2
    f32x4 acc0[4][2];
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc0[T][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        __builtin_amdgcn_s_barrier();            // barrier(nk1 + 2)
        asm volatile("" ::: "memory");
        const char* sb = db_sm + ((nk1 + 2) % DB_NST) * DB_STAGE_B;
#pragma unroll
        for (int p = 0; p < 4; ++p) {            // k block p >> 1 (rows + 64), chunk (p & 1) * 4 + q
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const bf16x8 wf = wfrag(sb, (p >> 1) * 64, T, (p & 1) * 4 + q);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc0[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, db_frag(h1f[j][p]), acc0[T][j], 0, 0, 0);
            }
        }
    }
    __builtin_amdgcn_s_barrier();                // the loaders' closing barrier
    // lane holds columns 32 (T >> 1) + 8 q + 4 (T & 1) + r of row (j, r16): fp32, 4-byte stores (A = 50 is not a multiple of 4 and the two
    // floats behind the attr block of a record row belong to other gradients)
    const __amdgpu_buffer_rsrc_t rz = buf_rsrc(a.dza);
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = (T >> 1) * 32 + q * 8 + (T & 1) * 4 + r;
                buf_store4(rz, (rok[j] && col < a.A) ? (grow[j] * (unsigned)a.ld_dza + (unsigned)col) * 4u : BUF_OOB, __float_as_uint(acc0[T][j][r]));
            }
}

}  // namespace

bool dec_fused_bwd_supported(int A, int n_out, long long N, int ld_s, int ld2, int ld_dza) {
    return A >= 1 && A <= 64 && n_out >= 64 && (n_out & 7) == 0 && (ld_s & 7) == 0 && (ld2 & 7) == 0 && N > 0 && N * (long long)ld_s * 2 < (1ll << 32) &&
           N * (long long)DB_H2 * 2 < (1ll << 32) && N * (long long)ld_dza * 4 < (1ll << 32) && 256ll * ld2 * 2 < (1ll << 32);
}

// dL: bf16 [N][ld_s] d-logits (n_out columns); W2t / W1t / W0t: the transposed bf16 weights [256][ld2], [128][256], [A][128] (k_prep mode 1);
// H2 / H1: the stored forward activations (relu gates); dH2 / dH1 (bf16) and dza (fp32 [N][ld_dza], A columns) are overwritten.
int dec_fused_bwd(const void* dL, int ld_s, const void* W2t, int ld2, const void* W1t, const void* W0t, const void* H2, const void* H1, void* dH2,
                  void* dH1, float* dza, int ld_dza, long long N, int A, int n_out, hipStream_t s) {
    if (!dec_fused_bwd_supported(A, n_out, N, ld_s, ld2, ld_dza)) return SPAIR_ERR_UNSUPPORTED;
    DecBwdArgs a;
    a.dL = reinterpret_cast<const u16*>(dL); a.ld_s = ld_s;
    a.W2t = reinterpret_cast<const u16*>(W2t); a.ld2 = ld2;
    a.W1t = reinterpret_cast<const u16*>(W1t);
    a.W0t = reinterpret_cast<const u16*>(W0t); a.A = A;
    a.H2 = reinterpret_cast<const u16*>(H2); a.H1 = reinterpret_cast<const u16*>(H1);
    a.dH2 = reinterpret_cast<u16*>(dH2); a.dH1 = reinterpret_cast<u16*>(dH1);
    a.dza = dza; a.ld_dza = ld_dza; a.N = (int)N; a.K1 = n_out;
    constexpr int lds = DB_NST * DB_STAGE_B;      // 96 KB
    static std::atomic<unsigned long long> attr_done{0};
    if (spair_dyn_lds_once(reinterpret_cast<const void*>(&k_dec_bwd), lds, attr_done) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_dec_bwd, dim3((unsigned)((N + DB_ROWS - 1) / DB_ROWS)), dim3(512), lds, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// unit-level C ABI (tests)
extern "C" int spair_decoder_bwd16(const void* dlogits16, int ld_s, const void* W2t16, int ld2, const void* W1t16, const void* W0t16, const void* H2,
                                   const void* H1, void* dH2, void* dH1, float* d_z_attr, int ld_dza, long long N, int A, int n_out, void* stream) {
    return dec_fused_bwd(dlogits16, ld_s, W2t16, ld2, W1t16, W0t16, H2, H1, dH2, dH1, d_z_attr, ld_dza, N, A, n_out, (hipStream_t)stream);
}
