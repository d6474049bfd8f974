// Fused object decoder forward of the bf16 step (reference: models.py:474-492): z_attr -> 128 -> 256 -> P*P*2 logits -> (grey, alpha)
// sprites, ONE launch instead of three GEMM launches.
//
// "Activation-stationary": every wave keeps ITS 64 rows' activations in registers as MFMA B-operand fragments through all three layers
// and only weights move.  All products are computed TRANSPOSED, C'[n][row] = sum_k W[n][k] . act[row][k] (weights are the A operand,
// activations the B operand of v_mfma_f32_16x16x32_bf16): the accumulator layout of C' (lane group q holds output features 4q..4q+3 of a
// 16-feature tile for row lane&15) is, up to a permutation of the FEATURE order inside a pair of tiles, exactly the B-operand layout of the
// next layer (lane group q holds k = 8q..8q+7 for row lane&15).  That permutation is folded into the order in which the weight ROWS of
// each tile are packed (k_dec_pack below): tile 2p row (q, r) holds feature 32p + 8q + r, tile 2p+1 row (q, r) feature 32p + 8q + 4 + r,
// so that after the two tiles of a pair a lane owns features 32p + 8q .. 32p + 8q + 7 of its row -- the next layer's k order is the
// natural one, and the 8 values are also one contiguous 16-byte piece of the row-major activation / sprite row in HBM.  Nothing goes
// through LDS except the weights.
//
// Weights arrive as a linear STREAM of 8-KiB slots (8 fragments of 1 KiB, lane l's 16 bytes at l*16: what ds_read_b128 wants,
// conflict-free) that the pack kernel lays out in consumption order; the four waves of a workgroup copy it L2 -> LDS with
// global_load_lds_dwordx4 into a ring of 5 slots, 4 slots (32 KiB per workgroup) in flight behind a COUNTED s_waitcnt vmcnt and one raw
// s_barrier per slot (no vmcnt(0) in the loop; stores are counted in: gfx9 has one in-order counter for loads and stores).
//
// Work split: a workgroup = 256 rows x one HALF of the 1568 output columns (the two small layers are recomputed by both halves: 9 % of
// the flops, and it doubles the number of workgroups to 2 per CU); only half 0 stores the hidden activations (the weight-gradient GEMMs
// and the relu gates of the backward read them).
#include <type_traits>
#include "common.h"
#include "layout.h"
#include "dec_fused.h"

namespace {
constexpr float DF_L2E = 1.4426950408889634f;

constexpr int DF_WAVES = 4, DF_RT = 4;                 // waves per workgroup, 16-row tiles per wave
constexpr int DF_ROWS_W = 16 * DF_RT, DF_ROWS = DF_ROWS_W * DF_WAVES;
constexpr int DF_SLOT_B = 8 * 1024;                    // bytes per slot = 8 fragments
constexpr int DF_D = 4, DF_NS = DF_D + 1;              // slots in flight, ring size
constexpr int DF_H1 = SP_DEC_H1, DF_H2 = SP_DEC_H2;    // 128, 256
constexpr int DF_K0 = 64;                              // z_attr columns padded to two k-steps
constexpr int DF_SLOTS_SMALL = 2 * (2 + 8);            // W0 (2 slots) + W1 (8 slots), once per pair of row tiles
constexpr int DF_MAX_PAIRS_HALF = 32;

struct DecFwdArgs {
    const __bf16* Za; int ld_za;           // [N][ld_za] bf16, columns >= A are finite (their weights are zero)
    const uint4* stream[2]; int n_slots[2]; int pair0[2];
    const float* b0; const float* b1; const float* b2;
    __bf16* H1; __bf16* H2; _Float16* S; int ld_s;
    int N, n_out;
    float obj_scale, alpha_scale, alpha_bias;
};

// VM operations (stores) a wave issues while it consumes stream slot i -- the counted waits below are exact because of this table.
__host__ __device__ constexpr int df_stores_of_slot(int i) {
    // per pair of row tiles: 2 W0 slots (2 feature pairs each x 2 row tiles = 4 stores), 8 W1 slots (1 feature pair x 2 row tiles = 2 stores)
    return i < 0 ? 0 : (i < DF_SLOTS_SMALL ? ((i % 10) < 2 ? 4 : 2) : (((i - DF_SLOTS_SMALL) & 1) ? DF_RT : 0));
}
// operations issued AFTER the two DMA instructions of slot i and before the wait at the top of iteration i: the DMAs of slots i+1 .. i+D-1
// and the stores of iterations i-D .. i-1
__host__ __device__ constexpr int df_wait_of_slot(int i) {
    int n = 2 * (DF_D - 1);
    for (int k = i - DF_D; k < i; ++k) n += df_stores_of_slot(k);
    return n;
}

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint4 pack_relu_bf16(const f32x4& a, const f32x4& b) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)fmaxf(a[e], 0.f); o[4 + e] = (__bf16)fmaxf(b[e], 0.f); }
    return *reinterpret_cast<uint4*>(&o);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const uint4& v) {
    union { uint4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

template <int W>
__device__ __forceinline__ void df_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
}

__global__ __launch_bounds__(DF_WAVES * 64, 2) void k_dec_fwd(DecFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char df_sm[];      // ring [DF_NS][8 KiB] | biases: b0 [128] b1 [256] b2 (this half) [<= 1024]
    char* ring = df_sm;
    float* bias_sh = reinterpret_cast<float*>(df_sm + DF_NS * DF_SLOT_B);
    // The biases are READ through the same 16-byte bf16 vector type as the weight fragments: hipcc's wait insertion puts an s_waitcnt vmcnt(0)
    // in front of every LDS load it cannot tell apart from the pending LDS-DMA writes (it decides by type-based alias info), which drained
    // the ring before each bias read when they were float4 loads.
    auto bias4 = [&](int idx) -> float4 {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ring + DF_NS * DF_SLOT_B + idx * 4);
        float4 o;
        __builtin_memcpy(&o, &v, 16);
        return o;
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;
    const int half = blockIdx.x & 1, mt = blockIdx.x >> 1;
    const int row0 = mt * DF_ROWS + wave * DF_ROWS_W;
    const int ns = a.n_slots[half], pair0 = a.pair0[half], npairs = (ns - DF_SLOTS_SMALL) >> 1;
    const char* strm = reinterpret_cast<const char*>(a.stream[half]);

    // ---- the rows' z_attr as B fragments, before any DMA is in flight (an ordinary load's use drains the queue: vmcnt(0))
    uint4 za[DF_RT][2];
    {
        const __amdgpu_buffer_rsrc_t rz = buf_rsrc(a.Za);
#pragma unroll
        for (int j = 0; j < DF_RT; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = row0 + j * 16 + r16, k = ks * 32 + q * 8;
                za[j][ks] = buf_load16(rz, (row < a.N && k + 8 <= a.ld_za) ? ((unsigned)row * (unsigned)a.ld_za + (unsigned)k) * 2u : BUF_OOB);
            }
    }
    for (int i = tid; i < DF_H1; i += DF_WAVES * 64) bias_sh[i] = a.b0[i];
    for (int i = tid; i < DF_H2; i += DF_WAVES * 64) bias_sh[DF_H1 + i] = a.b1[i];
    // decoder.out's weights arrive pre-multiplied by -scale * log2(e) (k_dec_pack), so the bias is folded the same way and the accumulator
    // is directly the exp2 argument of the analytic sigmoid: even columns grey (obj scale), odd alpha (alpha scale and bias)
    for (int i = tid; i < npairs * 32; i += DF_WAVES * 64) {
        const float bv = (pair0 * 32 + i) < a.n_out ? a.b2[pair0 * 32 + i] : 0.f;
        bias_sh[DF_H1 + DF_H2 + i] = (i & 1) ? -(bv * a.alpha_scale + a.alpha_bias) * DF_L2E : -(bv * a.obj_scale) * DF_L2E;
    }
    // "use" the loaded fragments here, where nothing else is in flight: the compiler's own wait for them then sits in front of the first DMA
    // instead of in front of the first MFMA (where it would be an s_waitcnt vmcnt(0) that drains the prologue's four slots)
#pragma unroll
    for (int j = 0; j < DF_RT; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(za[j][ks].x), "+v"(za[j][ks].y), "+v"(za[j][ks].z), "+v"(za[j][ks].w));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto issue = [&](int s) {         // slot s of the stream -> ring slot s % DF_NS; past the end: the last slot again (never consumed) so that
                                      // every iteration issues the same number of operations
        const char* src = strm + (size_t)min(s, ns - 1) * DF_SLOT_B + (wave * 2) * 1024 + lane * 16;
        char* dst = ring + (s % DF_NS) * DF_SLOT_B + (wave * 2) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024),
                                         (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < DF_D; ++s) issue(s);

    // top of iteration i: slot i has landed for every wave, slot i-1's buffer is free -> refill it with slot i+D
#define DF_ACQUIRE(i_const, i_run)                     \
    do {                                               \
        df_wait<df_wait_of_slot(i_const)>();           \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
        issue((i_run) + DF_D);                         \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)
    auto frag = [&](int slot_i, int f) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(ring + (slot_i % DF_NS) * DF_SLOT_B + f * 1024 + lane * 16);
    };

    const __amdgpu_buffer_rsrc_t rh1 = buf_rsrc(a.H1), rh2 = buf_rsrc(a.H2), rs = buf_rsrc(a.S);
    const bool st_hidden = half == 0;
    uint4 h2f[DF_RT][8];               // the wave's 64 rows x 256 hidden features: B fragments of decoder.out (128 registers)

    // ---- layers 0 and 1, two row tiles at a time (all four at once would need 64 + 128 + 32 registers of fragments before any scratch)
#pragma unroll
    for (int jp = 0; jp < DF_RT / 2; ++jp) {
        uint4 h1f[2][4];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {                         // W0: slot = feature pairs 2 sl, 2 sl + 1; fragment (pl, ks, t) at (pl * 2 + ks) * 2 + t
            constexpr int dummy = 0; (void)dummy;
            const int i_run = jp * 10 + sl;
            if (jp == 0 && sl == 0) DF_ACQUIRE(0, i_run); else if (jp == 0) DF_ACQUIRE(1, i_run); else if (sl == 0) DF_ACQUIRE(10, i_run); else DF_ACQUIRE(11, i_run);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int p = sl * 2 + pl;
                const float4 bA = bias4(p * 32 + q * 8), bB = bias4(p * 32 + q * 8 + 4);
                f32x4 acc[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) { acc[0][jj] = (f32x4){bA.x, bA.y, bA.z, bA.w}; acc[1][jj] = (f32x4){bB.x, bB.y, bB.z, bB.w}; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 w0 = frag(i_run, (pl * 2 + ks) * 2), w1 = frag(i_run, (pl * 2 + ks) * 2 + 1);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        acc[0][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, as_bf16x8(za[jp * 2 + jj][ks]), acc[0][jj], 0, 0, 0);
                        acc[1][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, as_bf16x8(za[jp * 2 + jj][ks]), acc[1][jj], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    h1f[jj][p] = pack_relu_bf16(acc[0][jj], acc[1][jj]);
                    const int row = row0 + (jp * 2 + jj) * 16 + r16;
                    buf_store16(rh1, (st_hidden && row < a.N) ? ((unsigned)row * DF_H1 + p * 32 + q * 8) * 2u : BUF_OOB, h1f[jj][p]);
                }
            }
        }
#pragma unroll
        for (int p2 = 0; p2 < 8; ++p2) {                         // W1: one slot per feature pair; fragment (ks, t) at ks * 2 + t
            const int i_run = jp * 10 + 2 + p2;
            // (the wait count only depends on the position inside the 10-slot group except for the first group's first slots)
            if (jp == 0) {
                switch (p2) { case 0: DF_ACQUIRE(2, i_run); break; case 1: DF_ACQUIRE(3, i_run); break; case 2: DF_ACQUIRE(4, i_run); break;
                              case 3: DF_ACQUIRE(5, i_run); break; default: DF_ACQUIRE(6, i_run); break; }
            } else {
                switch (p2) { case 0: DF_ACQUIRE(12, i_run); break; case 1: DF_ACQUIRE(13, i_run); break; case 2: DF_ACQUIRE(14, i_run); break;
                              case 3: DF_ACQUIRE(15, i_run); break; default: DF_ACQUIRE(16, i_run); break; }
            }
            const float4 bA = bias4(DF_H1 + p2 * 32 + q * 8), bB = bias4(DF_H1 + p2 * 32 + q * 8 + 4);
            f32x4 acc[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) { acc[0][jj] = (f32x4){bA.x, bA.y, bA.z, bA.w}; acc[1][jj] = (f32x4){bB.x, bB.y, bB.z, bB.w}; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 w0 = frag(i_run, ks * 2), w1 = frag(i_run, ks * 2 + 1);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    acc[0][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, as_bf16x8(h1f[jj][ks]), acc[0][jj], 0, 0, 0);
                    acc[1][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, as_bf16x8(h1f[jj][ks]), acc[1][jj], 0, 0, 0);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                h2f[jp * 2 + jj][p2] = pack_relu_bf16(acc[0][jj], acc[1][jj]);
                const int row = row0 + (jp * 2 + jj) * 16 + r16;
                buf_store16(rh2, (st_hidden && row < a.N) ? ((unsigned)row * DF_H2 + p2 * 32 + q * 8) * 2u : BUF_OOB, h2f[jp * 2 + jj][p2]);
            }
        }
    }

    // ---- decoder.out + sprite epilogue: per pair of 16-column tiles two slots (k-steps 0..3, 4..7; fragment (ks, t) at (ks & 3) * 2 + t)
    // The fragments of slot i+1 are read from LDS while the MFMAs of slot i run: at the top of iteration i the wave waits for slot i+1
    // (not i), and the barrier there also tells everyone that slot i's fragments are in registers -- its buffer is refilled at once.
    // Without this every slot started with a full LDS round trip (8 reads, s_waitcnt lgkmcnt(0)) in front of its 32 MFMAs.
    auto load_frags = [&](bf16x8 (&F)[8], int slot_i) {
#pragma unroll
        for (int f = 0; f < 8; ++f) F[f] = frag(slot_i, f);
    };
    bf16x8 F0[8], F1[8];
    {   // transition: the ring is one slot deeper from here on (slots i+1 .. i+D in flight at the top of iteration i)
        df_wait<df_wait_of_slot(DF_SLOTS_SMALL)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(DF_SLOTS_SMALL + DF_D);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(F0, DF_SLOTS_SMALL);
    }
    auto half_body = [&](int i_run, auto w_c, bf16x8 (&Fc)[8], bf16x8 (&Fn)[8], f32x4 (&acc)[2][DF_RT], int hs) {
        constexpr int W = decltype(w_c)::value;
        // this wave's reads of slot i_run are done: its buffer may be overwritten after the barrier.  (The builtin: hipcc's own wait insertion
        // does not see through an inline-asm s_waitcnt and would wait for these reads again, behind the next slot's.)
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        df_wait<W>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(i_run + 1 + DF_D);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(Fn, i_run + 1);
        __builtin_amdgcn_sched_barrier(0);       // the reads are ISSUED here (left alone hipcc sinks them below the MFMAs, right in front of the wait)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int j = 0; j < DF_RT; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fc[kq * 2], as_bf16x8(h2f[j][hs * 4 + kq]), acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Fc[kq * 2 + 1], as_bf16x8(h2f[j][hs * 4 + kq]), acc[1][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pair_body = [&](int pr, auto wa_c, auto wb_c) {
        const int i_run = DF_SLOTS_SMALL + 2 * pr;
        const float4 bA = bias4(DF_H1 + DF_H2 + pr * 32 + q * 8), bB = bias4(DF_H1 + DF_H2 + pr * 32 + q * 8 + 4);
        f32x4 acc[2][DF_RT];
#pragma unroll
        for (int j = 0; j < DF_RT; ++j) { acc[0][j] = (f32x4){bA.x, bA.y, bA.z, bA.w}; acc[1][j] = (f32x4){bB.x, bB.y, bB.z, bB.w}; }
        half_body(i_run, wa_c, F0, F1, acc, 0);
        half_body(i_run + 1, wb_c, F1, F0, acc, 1);
        const int col = (pair0 + pr) * 32 + q * 8;
#pragma unroll
        for (int j = 0; j < DF_RT; ++j) {
            h8_t o;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = acc[t][j][r];
                    o[t * 4 + r] = (_Float16)__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(u) + 1.f);
                }
            const int row = row0 + j * 16 + r16;
            uint4 ov;
            __builtin_memcpy(&ov, &o, 16);
            buf_store16(rs, (row < a.N && col < a.n_out) ? ((unsigned)row * (unsigned)a.ld_s + (unsigned)col) * 2u : BUF_OOB, ov);
        }
    };
    // waits (operations younger than the DMAs of slot i+1 at the top of iteration i): 6 DMAs + the stores of iterations i-4 .. i-1, except
    // that the transition step stored nothing -- pairs 0 and 1: (12, 10), from pair 2 on (14, 14)
    constexpr int S0 = DF_SLOTS_SMALL;
    constexpr int W20 = 6 + df_stores_of_slot(S0 - 3) + df_stores_of_slot(S0 - 2) + df_stores_of_slot(S0 - 1);
    constexpr int W21 = 6 + df_stores_of_slot(S0 - 2) + df_stores_of_slot(S0 - 1) + df_stores_of_slot(S0);
    constexpr int W22 = 6 + df_stores_of_slot(S0 - 1) + df_stores_of_slot(S0) + df_stores_of_slot(S0 + 1);
    constexpr int W23 = 6 + df_stores_of_slot(S0) + df_stores_of_slot(S0 + 1) + df_stores_of_slot(S0 + 2);
    pair_body(0, std::integral_constant<int, W20>{}, std::integral_constant<int, W21>{});
    if (npairs > 1) pair_body(1, std::integral_constant<int, W22>{}, std::integral_constant<int, W23>{});
#pragma unroll 1
    for (int pr = 2; pr < npairs; ++pr)
        pair_body(pr, std::integral_constant<int, df_wait_of_slot(S0 + 4)>{}, std::integral_constant<int, df_wait_of_slot(S0 + 5)>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail DMAs target this workgroup's LDS: they must land before it is released
#undef DF_ACQUIRE
}

// One thread per 16-byte lane piece of a fragment.  Stream layout per half (slots of 8 fragments):
//   [W0: 2 slots][W1: 8 slots][W0][W1] (one group per pair of row tiles) then 2 slots per pair of 16-column tiles of this half's decoder.out columns
__global__ __launch_bounds__(256) void k_dec_pack(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2, int A,
                                                  int n_out, uint4* __restrict__ st0, uint4* __restrict__ st1, int ns0, int ns1, int pair0_1,
                                                  float obj_scale, float alpha_scale) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per0 = (long long)ns0 * 8 * 64, per1 = (long long)ns1 * 8 * 64;
    if (idx >= per0 + per1) return;
    const int half = idx >= per0;
    const long long li = half ? idx - per0 : idx;
    const int lane = (int)(li & 63), f = (int)((li >> 6) & 7), s = (int)(li >> 9);
    const int q = lane >> 4, m = lane & 15;
    const float* W; int out, in, T, ks;
    bool last = false;
    if (s < DF_SLOTS_SMALL) {
        const int g = s % 10;
        if (g < 2) { W = W0; out = DF_H1; in = A; const int pl = f >> 2; ks = (f >> 1) & 1; T = 2 * (g * 2 + pl) + (f & 1); }
        else { W = W1; out = DF_H2; in = DF_H1; ks = f >> 1; T = 2 * (g - 2) + (f & 1); }
    } else {
        const int u = s - DF_SLOTS_SMALL;
        last = true;
        W = W2; out = n_out; in = DF_H2; ks = (u & 1) * 4 + (f >> 1); T = 2 * ((half ? pair0_1 : 0) + (u >> 1)) + (f & 1);
    }
    const int n = (T >> 1) * 32 + (m >> 2) * 8 + (T & 1) * 4 + (m & 3), k0 = ks * 32 + q * 8;
    const float sc = last ? -((n & 1) ? alpha_scale : obj_scale) * DF_L2E : 1.f;      // sigmoid(scale * v + bias) = 1 / (1 + exp2(acc))
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((n < out && k0 + e < in) ? W[(size_t)n * in + k0 + e] * sc : 0.f);
    (half ? st1 : st0)[li] = *reinterpret_cast<uint4*>(&o);
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool dec_fused_supported(int A, int n_out, int ld_za, long long N, int ld_s) {
    return A <= DF_K0 && (n_out % 32) == 0 && n_out / 32 <= 2 * DF_MAX_PAIRS_HALF && n_out >= 64 && (ld_za & 7) == 0 && (ld_s & 7) == 0 &&
           N * (long long)ld_s * 2 < (1ll << 32) && N * (long long)DF_H2 * 2 < (1ll << 32);
}
static void dec_split(int n_out, int& np0, int& np1) { const int np = n_out / 32; np0 = (np + 1) / 2; np1 = np - np0; }
size_t dec_fused_stream_bytes(int n_out) {
    int np0, np1;
    dec_split(n_out, np0, np1);
    return (size_t)(2 * DF_SLOTS_SMALL + 2 * (np0 + np1)) * DF_SLOT_B;
}
// stream: dec_fused_stream_bytes(n_out) bytes of workspace; W0 [128][A], W1 [256][128], W2 [n_out][256] fp32 row-major (the parameters)
int dec_fused_pack(const float* W0, const float* W1, const float* W2, int A, int n_out, float obj_scale, float alpha_scale, void* stream_buf,
                   hipStream_t s) {
    int np0, np1;
    dec_split(n_out, np0, np1);
    const int ns0 = DF_SLOTS_SMALL + 2 * np0, ns1 = DF_SLOTS_SMALL + 2 * np1;
    uint4* st0 = reinterpret_cast<uint4*>(stream_buf);
    uint4* st1 = st0 + (size_t)ns0 * (DF_SLOT_B / 16);
    const long long total = (long long)(ns0 + ns1) * 8 * 64;
    hipLaunchKernelGGL(k_dec_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W0, W1, W2, A, n_out, st0, st1, ns0, ns1, np0,
                       obj_scale, alpha_scale);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
int dec_fused_fwd(const void* Za16, int ld_za, const void* stream_buf, const float* b0, const float* b1, const float* b2, void* H1, void* H2,
                  void* S, int ld_s, long long N, int A, int n_out, float obj_scale, float alpha_scale, float alpha_bias, hipStream_t s) {
    if (!dec_fused_supported(A, n_out, ld_za, N, ld_s)) return SPAIR_ERR_UNSUPPORTED;
    int np0, np1;
    dec_split(n_out, np0, np1);
    DecFwdArgs a;
    a.Za = reinterpret_cast<const __bf16*>(Za16); a.ld_za = ld_za;
    a.n_slots[0] = DF_SLOTS_SMALL + 2 * np0; a.n_slots[1] = DF_SLOTS_SMALL + 2 * np1; a.pair0[0] = 0; a.pair0[1] = np0;
    a.stream[0] = reinterpret_cast<const uint4*>(stream_buf);
    a.stream[1] = a.stream[0] + (size_t)a.n_slots[0] * (DF_SLOT_B / 16);
    a.b0 = b0; a.b1 = b1; a.b2 = b2;
    a.H1 = reinterpret_cast<__bf16*>(H1); a.H2 = reinterpret_cast<__bf16*>(H2); a.S = reinterpret_cast<_Float16*>(S); a.ld_s = ld_s;
    a.N = (int)N; a.n_out = n_out; a.obj_scale = obj_scale; a.alpha_scale = alpha_scale; a.alpha_bias = alpha_bias;
    const size_t lds = (size_t)DF_NS * DF_SLOT_B + (size_t)(DF_H1 + DF_H2 + DF_MAX_PAIRS_HALF * 32) * 4;
    const unsigned mtiles = (unsigned)((N + DF_ROWS - 1) / DF_ROWS);
    hipLaunchKernelGGL(k_dec_fwd, dim3(2 * mtiles), dim3(DF_WAVES * 64), lds, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

// unit-level C ABI (tests): packs the weights into `stream_buf` (dec_fused_stream_bytes) and runs the fused forward
extern "C" int64_t spair_decoder_fwd16_scratch_bytes(int n_out) { return (int64_t)dec_fused_stream_bytes(n_out); }
extern "C" int spair_decoder_fwd16(const void* z_attr16, int ld_za, const float* W0, const float* b0, const float* W1, const float* b1,
                                   const float* W2, const float* b2, void* H1, void* H2, void* sprites, int ld_s, long long N, int A, int n_out,
                                   float obj_scale, float alpha_scale, float alpha_bias, void* stream_buf, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!dec_fused_supported(A, n_out, ld_za, N, ld_s)) return SPAIR_ERR_UNSUPPORTED;
    int rc = dec_fused_pack(W0, W1, W2, A, n_out, obj_scale, alpha_scale, stream_buf, s);
    if (rc != SPAIR_OK) return rc;
    return dec_fused_fwd(z_attr16, ld_za, stream_buf, b0, b1, b2, H1, H2, sprites, ld_s, N, A, n_out, obj_scale, alpha_scale, alpha_bias, s);
}
