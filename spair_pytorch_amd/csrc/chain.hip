// Persistent fused kernels for the sequential per-cell encoder (reference: models.py:68-117).
//
// MI355X-first design of the one truly sequential part of SPAIR: samples are independent, so ONE
// WORKGROUP OWNS ONE SAMPLE and walks all 3G-2 dependency wavefronts by itself -- no inter-workgroup
// synchronisation, no kernel boundary per layer (the per-wavefront path needs ~18 launches x 46 steps).
// Per wavefront the <=16 independent cells of the sample form one 16-row MFMA tile; activations,
// context records and latents stay in LDS; the ~0.9 MB of bf16 weights are streamed from L2 every
// step in MFMA-FRAGMENT-PACKED order (one contiguous 1 KiB wave-load per 16x32 operand, prepared by
// k_prep mode 4) straight into registers -- they are used once per step and not shared between
// waves, so LDS staging would be pure overhead (cdna_hip_programming.md §5, "GEMV / M <= 16" row).
// Everything the backward pass and the weight-gradient GEMMs need is written to the same HBM row
// buffers the per-wavefront path produces, so the two paths are interchangeable (and are compared
// against each other in tests/test_chain_gpu.py).
#include <type_traits>
#include "cell_math.h"
#include "chain.h"
#include "stn_math.h"

namespace {

constexpr int MT = 16;                    // rows per wavefront tile
constexpr int F = 100, REC = 56, CTX = 224, A_ = 50, NP = 100, GLN = 784, PG = 28;    // PG: glimpse side (chain_fwd_supported requires P == 28)
// LDS row pitches of the MFMA A operands: K + 16 elements, i.e. a multiple of 64 B plus 32.  ds_read_b128 serves a wave in four groups of 16
// lanes (rows {0-3, 12-15} of one k-chunk with rows {4-11} of the next) against 64 banks: with the usual K + 8 (pitch = 16 mod 64 B) two rows of
// every group share a bank quad and each fragment read takes 8 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.47 fwd, 0.35 bwd).
constexpr int CH_PAD = 16;                // (same box, same run: K + 8 -> K + 16 = chain fwd 0.833 -> 0.820 ms, bwd 0.865 -> 0.852)
constexpr int KC = 352, LD_XC = KC + CH_PAD;  // [feat | ctx] padded to 11 k-steps
constexpr int KX = 160, LD_XT = KX + CH_PAD;  // [pass | box | attr | depth] padded to 5 k-steps
constexpr int KG = 800, LD_GL = KG + CH_PAD;  // glimpse padded to 25 k-steps
constexpr int LD_H = 256 + CH_PAD;
constexpr int LD_O = 112;

// Workgroup barrier that orders LDS only.  __syncthreads() also drains every outstanding global access (s_waitcnt vmcnt(0)):
// here that would stall each of the ~20 stages per wavefront on the acks of its activation stores and would force the weight
// fragments prefetched for the next layer to land before the barrier.  Nothing written to HBM by these kernels is read back by
// them, so LDS ordering is all the stages need (cdna_hip_programming.md §5 "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// orders a wave's own LDS writes before its later LDS reads (other lanes' data), no workgroup barrier
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bf16x8 as_frag(const uint4& v) {
    union { uint4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// Weight-fragment pipeline.  Every wave owns ONE 16-column tile per (half-)layer; its share of the layer is the linear sequence
// of that tile's KT 1-KiB fragments, RD of which are kept in flight in a register ring.  The code between a fragment's load and
// its MFMA is STRAIGHT-LINE: waves without a tile (NT not a multiple of 8) run on a clamped tile index and only skip the store --
// a wave-uniform `if (nt < NT)` around the loads/MFMAs splits basic blocks and makes hipcc wait vmcnt(0) before every MFMA,
// which serialises the whole stream (measured: ~4.5k cycles per stage regardless of its size).  The ring lives across layer
// boundaries and barriers: pipe_fill() for layer l+1 is issued right after the MFMAs of layer l.
constexpr int NW = 8;                     // waves per workgroup
// band split: polls of the neighbouring band's counter before a wait is abandoned (~0.3 s; a legitimate wait is the other band's head start,
// half a millisecond) -- a kernel of this library never spins without a bound
constexpr int CHAIN_SPIN_LIMIT = 1 << 18;
constexpr int NTH = NW * 64;
constexpr int ENC0_RES = 7;   // k-steps of encoder layer 0 (of 25) whose fragments are resident in the forward kernel's registers
constexpr int RD = 12;      // weight fragments in flight per wave (measured: 16 / 20 / 24 change nothing, 0.91 -> 0.93-0.95 ms)
// the row-buffer stores are non-temporal: they stream past the L2 that holds the weights every workgroup re-reads each wavefront (chain fwd
// 0.854 -> 0.836 ms, bwd 0.861 -> 0.851)
template <typename T> __device__ __forceinline__ void ch_gstore_nt(T* p, const T& v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void ch_gstore_nt(float4* p, const float4& v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(p));
}
#define CH_GSTORE(p, v) ch_gstore_nt((p), (v))
// the backward kernel's read-once row data (bundle, glimpse derivatives) is loaded non-temporally (chain bwd 0.852 -> 0.833 ms)
__device__ __forceinline__ uint4 ch_gload16(const void* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}
#define CH_GLOAD16(p) ch_gload16(p)
__device__ __forceinline__ float4 ch_gloadf4(const float* p) {
    const uint4 v = ch_gload16(p);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#define CH_GLOADF4(p) ch_gloadf4(p)
struct WPipe { uint4 q[RD]; };

template <int KT, int NT, int NT0>
__device__ __forceinline__ const uint4* tile_base(const uint4* __restrict__ Wp, int wave, int lane) {
    const int nt = min(NT0 + wave, NT - 1);
    return Wp + (size_t)nt * KT * 64 + lane;
}

template <int KT, int NT, int NT0 = 0>
__device__ __forceinline__ void pipe_fill(const uint4* __restrict__ Wp, WPipe& p, int wave, int lane) {
    const uint4* base = tile_base<KT, NT, NT0>(Wp, wave, lane);
#pragma unroll
    for (int f = 0; f < RD; ++f)
        if (f < KT) p.q[f] = base[f * 64];
}

// relu layers leave their sign bits as wave ballots (see wg_store_t): the backward kernel reads 2 KB of bits per wavefront instead of
// re-reading 56 KB of activations
constexpr int MB_HB1 = 0, MB_HB2 = 7, MB_HE1 = 14, MB_HE2 = 30, MB_HZ1 = 38, MB_HZ2 = 45, MB_HO1 = 52, MB_HO2 = 59, MB_TILES = 66;
// The "store wave": layers with 7 column tiles leave the 8th wave without work.  It copies the PREVIOUS stage's output rows from
// LDS to their HBM row buffer (coalesced 16-byte stores) while the other seven compute -- their instruction streams then contain no
// global store at all, so a wait for a weight fragment is only ever a wait for weight fragments.  (Hidden activations reach HBM as
// the bf16-rounded values the next layer consumed; the weight-gradient GEMMs round their operands to bf16 anyway.)
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 o;
    o[0] = (__bf16)a; o[1] = (__bf16)b; o[2] = (__bf16)c; o[3] = (__bf16)d;
    return o;
}

// ---- chain-local fast math.  The latent transforms sit on the critical path of a 46-step dependent chain as ~120-instruction sequences of
// one or four waves (IEEE division = 10 instructions, expf = 12): here they are v_exp_f32 / v_rcp_f32 / v_log_f32 forms, ~1 ulp each --
// far inside what the bf16 operands of the surrounding GEMMs do to the same values (the fp32 parity mode never runs these kernels).
__device__ __forceinline__ float ch_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896f * x)); }
__device__ __forceinline__ float ch_log(float x) { return 0.693147180559945f * __builtin_amdgcn_logf(x); }

// The store wave's row copier: PIECES pieces of PB (8 or 16) bytes per row, LDS -> HBM row buffer.  Eight lanes serve one row (a pass
// covers 8 rows: every row of a wavefront on grids up to 16 x 16), so the row's base addresses are formed once and every further piece
// is an instruction-immediate offset on both sides: 2 instructions per 64 pieces, all LDS reads of the pass issued before its first store
// (a serial read -> wait -> store loop with a division per piece made the store wave the slowest wave of the 7-tile stages).
template <int PIECES, int PB>
__device__ __forceinline__ void copy_rows8(const void* lds_, int lds_pitch, void* hbm_, unsigned hbm_pitch, const int* row_r, int r0, int nc, int lane) {
    constexpr int NI = (PIECES + 7) / 8;
    typedef unsigned int uv __attribute__((ext_vector_type(PB / 4)));
    const int row = r0 + (lane >> 3), p0 = lane & 7;
    const char* src = reinterpret_cast<const char*>(lds_) + row * lds_pitch + p0 * PB;
    char* dst = reinterpret_cast<char*>(hbm_) + ((unsigned)row_r[row] * hbm_pitch + (unsigned)(p0 * PB));      // every row buffer is < 4 GB
    uv v[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (8 * (i + 1) <= PIECES || p0 + 8 * i < PIECES) v[i] = *reinterpret_cast<const uv*>(src + i * 8 * PB);
    if (row < nc) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (8 * (i + 1) <= PIECES || p0 + 8 * i < PIECES) CH_GSTORE(reinterpret_cast<uv*>(dst + i * 8 * PB), v[i]);
    }
}
template <int PIECES, int PB>
__device__ __forceinline__ void copy_rows_w(const void* lds, int lds_pitch, void* hbm, size_t hbm_pitch, const int* row_r, int nc, int lane) {
    copy_rows8<PIECES, PB>(lds, lds_pitch, hbm, (unsigned)hbm_pitch, row_r, 0, nc, lane);
    if (nc > 8) copy_rows8<PIECES, PB>(lds, lds_pitch, hbm, (unsigned)hbm_pitch, row_r, 8, nc, lane);
}

// ---- lean GEMM stage, TRANSPOSED MFMA form (round 4) --------------------------------------------------------------------------------------
// The chain kernels are bound by the instruction streams of their ~16 dependent stages (ablation: with every weight load removed the forward
// kernel still takes 0.63 of its 0.84 ms; a 4-k-step layer ran ~140 instructions per wave at ~9 cycles each), not by the weight stream.  The
// stage is therefore written for instruction count and exposed latency:
//  * operands swapped -- the weight fragment is the A operand, the activation tile the B operand (the register layouts of the two are the
//    same, so the packs and the LDS reads do not change): D^T[col][row], a lane ends up with FOUR CONSECUTIVE COLUMNS of ONE row
//    (col = tile*16 + (lane>>4)*4 + r, row = lane & 15) -- one 8-byte LDS write of 4 bf16 (or one 16-byte write of 4 floats) instead of four
//    2-byte writes with four addresses;
//  * the accumulator starts as the bias quad (one 16-byte LDS read issued with the activation reads): no bias adds;
//  * every activation fragment of the stage is read before the first MFMA (the compiler otherwise re-used one register quad and exposed an
//    LDS round trip per k-step);
//  * relu sign bits: v_cmp leaves each ballot in an SGPR pair; lane 0 writes the tile's four words with two 16-byte stores.
// Sign-bit layout (shared with k_chain_bwd): word tile*4 + r, bit = lane  <->  column tile*16 + (lane>>4)*4 + r, row lane & 15.
template <int KT0, int KT1, int NT, int NT0 = 0>
__device__ __forceinline__ void wg_gemm_t(const __bf16* inA, int ldA, const __bf16* inB, int ldB, const uint4* __restrict__ Wp, WPipe& p,
                                          const float* bias_l, f32x4& acc, int wave, int lane) {
    constexpr int KT = KT0 + KT1;
    static_assert(KT <= 16, "all activation fragments of the stage are held in registers");
    const uint4* base = tile_base<KT, NT, NT0>(Wp, wave, lane);
    const int nt = min(NT0 + wave, NT - 1);
    const __bf16* pa = inA + (lane & 15) * ldA + (lane >> 4) * 8;
    const __bf16* pb = (KT1 > 0) ? inB + (lane & 15) * ldB + (lane >> 4) * 8 : pa;
    bf16x8 x[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
        x[kt] = *reinterpret_cast<const bf16x8*>((kt < KT0) ? pa + kt * 32 : pb + (kt - KT0) * 32);
    acc = *reinterpret_cast<const f32x4*>(bias_l + nt * 16 + (lane >> 4) * 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < KT; ++f) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[f % RD]), x[f], acc, 0, 0, 0);
        if (f + RD < KT) p.q[f % RD] = base[(size_t)(f + RD) * 64];
    }
}

// EARLY PRODUCTS (round 5).  Z0 and OBJ0 multiply [feat | ctx | pass | box | attr (| depth)]: the first KC = 352 columns (11 of their 16
// k-steps) have been in LDS since the wavefront's S0 stage, and every GEMM stage takes its pack bytes at ~105 GB/s of L1 fill (the two
// stages: 112 KiB = 1.05 us each).  Those 11 k-steps of BOTH layers (they share the activation fragments) are multiplied inside the glimpse
// sampling stage instead, which streams no packs: timing-only removal of them priced the move at 1.65 us of a 15.8-us wavefront.  The
// partial sums travel in two accumulators; the Z0 / OBJ0 stages start from them and run their last KTOT - KSKIP k-steps (wg_gemm_tk).
// Same products in the same order as the undivided stage: results are bit-identical.
template <int KTOT, int KSKIP, int NT>
__device__ __forceinline__ void pipe_fill_k(const uint4* __restrict__ Wp, WPipe& p, int wave, int lane) {
    const uint4* base = tile_base<KTOT, NT, 0>(Wp, wave, lane) + (size_t)KSKIP * 64;
#pragma unroll
    for (int f = 0; f < KTOT - KSKIP; ++f) p.q[f] = base[f * 64];
}
template <int KTOT, int KSKIP>
__device__ __forceinline__ void wg_gemm_tk(const __bf16* inB, int ldB, WPipe& p, f32x4& acc, int lane) {
    constexpr int KT = KTOT - KSKIP;
    static_assert(KT <= RD, "the remaining fragments sit in the ring");
    const __bf16* pb = inB + (lane & 15) * ldB + (lane >> 4) * 8;
    bf16x8 x[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) x[kt] = *reinterpret_cast<const bf16x8*>(pb + kt * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < KT; ++f) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[f]), x[f], acc, 0, 0, 0);
}
// ring <- k-steps [K0, K0 + N) of the two layers' tiles, slot 2 j = layer A, 2 j + 1 = layer B (N <= RD / 2)
template <int KTOT, int NT, int K0, int N>
__device__ __forceinline__ void early_fill(const uint4* __restrict__ Wa, const uint4* __restrict__ Wb, WPipe& p, int wave, int lane) {
    static_assert(2 * N <= RD, "two fragments per k-step");
    const uint4* ba = tile_base<KTOT, NT, 0>(Wa, wave, lane) + (size_t)K0 * 64;
    const uint4* bb = tile_base<KTOT, NT, 0>(Wb, wave, lane) + (size_t)K0 * 64;
#pragma unroll
    for (int j = 0; j < N; ++j) { p.q[2 * j] = ba[j * 64]; p.q[2 * j + 1] = bb[j * 64]; }
}
template <int N>
__device__ __forceinline__ void early_mfma(const bf16x8 (&x)[N], const WPipe& p, f32x4& accA, f32x4& accB) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[2 * j]), x[j], accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[2 * j + 1]), x[j], accB, 0, 0, 0);
    }
}
// stores through a buffer descriptor, non-temporal (see CH_GSTORE); a masked lane passes BUF_OOB
__device__ __forceinline__ void buf_store8_nt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned lo, unsigned hi) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{lo, hi}, r, (int)byte_off, 0, 2);
}
__device__ __forceinline__ void buf_store16_nt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, u32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 2);
}

// Two column tiles per wave (tiles `wave` and `wave + 8`: the encoder's 256-wide first layer) sharing every activation fragment; the K loop
// runs in chunks of CH k-steps, the next chunk's LDS reads issued before the current chunk's MFMAs.  Ring order: (k-step, tile) pairs.
template <int KT>
__device__ __forceinline__ void pipe_fill2(const uint4* __restrict__ Wp, WPipe& p, int wave, int lane) {
    const uint4* b0 = Wp + (size_t)wave * KT * 64 + lane;
    const uint4* b1 = Wp + (size_t)(wave + 8) * KT * 64 + lane;
#pragma unroll
    for (int s = 0; s < RD; ++s) p.q[s] = ((s & 1) ? b1 : b0)[(s >> 1) * 64];
}
// NRES: the last NRES k-steps of both tiles are RESIDENT -- 2 * NRES fragments this wave loaded once per kernel (`res`, the forward kernel's
// spare registers): they are not streamed again on any wavefront (the stage takes its pack bytes through the CU's L1 fill port).
template <int KT, int CH, int NRES>
__device__ __forceinline__ void wg_gemm_t2(const __bf16* in, int ld, const uint4* __restrict__ Wp, WPipe& p, const uint4 (&res)[2 * NRES + 1],
                                           const float* bias_l, f32x4& acc0, f32x4& acc1, int wave, int lane) {
    static_assert(KT % CH == 0 && RD % 2 == 0, "whole chunks; ring slots alternate between the two tiles");
    constexpr int NCH = KT / CH, AH = RD / 2;          // AH: k-steps the ring runs ahead
    constexpr int KS = KT - NRES;                      // k-steps that stream
    const uint4* b0 = Wp + (size_t)wave * KT * 64 + lane;
    const uint4* b1 = Wp + (size_t)(wave + 8) * KT * 64 + lane;
    const __bf16* pa = in + (lane & 15) * ld + (lane >> 4) * 8;
    bf16x8 x[2][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) x[0][j] = *reinterpret_cast<const bf16x8*>(pa + j * 32);
    acc0 = *reinterpret_cast<const f32x4*>(bias_l + wave * 16 + (lane >> 4) * 4);
    acc1 = *reinterpret_cast<const f32x4*>(bias_l + (wave + 8) * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < CH; ++j) x[(c + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(pa + ((c + 1) * CH + j) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int f = c * CH + j;
            if (f < KS) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[(2 * f) % RD]), x[c & 1][j], acc0, 0, 0, 0);
                if (f + AH < KS) p.q[(2 * f) % RD] = b0[(size_t)(f + AH) * 64];
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[(2 * f + 1) % RD]), x[c & 1][j], acc1, 0, 0, 0);
                if (f + AH < KS) p.q[(2 * f + 1) % RD] = b1[(size_t)(f + AH) * 64];
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(res[2 * (f - KS)]), x[c & 1][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(res[2 * (f - KS) + 1]), x[c & 1][j], acc1, 0, 0, 0);
            }
        }
    }
}

// SPLIT-bf16 stage (the box network, models.py:76-79): both operands as hi + lo bf16 pairs, three products per k-step (hi.hi + lo.hi + hi.lo;
// what is dropped is < 2^-16 of the term), fp32 accumulate -- the layer at nearly fp32 precision for twice the weight bytes of three small
// layers.  Why: the box places the glimpse and the sprite on the pixel grid; with plain bf16 operands the reconstruction moved by up to 0.05
// against the reference and the gradients of the box / encoder weights turned to cos 0.92 on the small fixtures (0.995 with this).
// Ring order: (k-step, {hi, lo}) pairs, as pipe_fill2 / wg_gemm_t2 with the two tiles replaced by the two packs of ONE tile.
template <int KT, int NT>
__device__ __forceinline__ void pipe_fill_s(const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo, WPipe& p, int wave, int lane) {
    const size_t o = (size_t)min(wave, NT - 1) * KT * 64 + lane;
#pragma unroll
    for (int s_ = 0; s_ < RD; ++s_)
        if ((s_ >> 1) < KT) p.q[s_] = ((s_ & 1) ? Wlo : Whi)[o + (s_ >> 1) * 64];
}
template <int KT, int NT, int CH>
__device__ __forceinline__ void wg_gemm_s(const __bf16* inHi, const __bf16* inLo, int ld, const uint4* __restrict__ Whi, const uint4* __restrict__ Wlo,
                                          WPipe& p, const float* bias_l, f32x4& acc, int wave, int lane) {
    constexpr int NCH = (KT + CH - 1) / CH, AH = RD / 2;
    const int nt = min(wave, NT - 1);
    const size_t o = (size_t)nt * KT * 64 + lane;
    const int eoff = (lane & 15) * ld + (lane >> 4) * 8;
    bf16x8 xh[2][CH], xl[2][CH];
#pragma unroll
    for (int j = 0; j < CH; ++j)
        if (j < KT) { xh[0][j] = *reinterpret_cast<const bf16x8*>(inHi + eoff + j * 32); xl[0][j] = *reinterpret_cast<const bf16x8*>(inLo + eoff + j * 32); }
    acc = *reinterpret_cast<const f32x4*>(bias_l + nt * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if ((c + 1) * CH + j < KT) {
                    xh[(c + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(inHi + eoff + ((c + 1) * CH + j) * 32);
                    xl[(c + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(inLo + eoff + ((c + 1) * CH + j) * 32);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int f = c * CH + j;
            if (f < KT) {
                const bf16x8 wh = as_frag(p.q[(2 * f) % RD]), wl = as_frag(p.q[(2 * f + 1) % RD]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[c & 1][j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[c & 1][j], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[c & 1][j], acc, 0, 0, 0);
                if (f + AH < KT) { p.q[(2 * f) % RD] = Whi[o + (size_t)(f + AH) * 64]; p.q[(2 * f + 1) % RD] = Wlo[o + (size_t)(f + AH) * 64]; }
            }
        }
    }
}

// epilogue of a lean stage.  ncol4 = number of output columns rounded up to 4 (quads past it are padding: not written); nbf = columns that
// get the bf16 LDS copy (the next layer's input tile); hbm16 = bf16 row buffer written straight from the epilogue (8 bytes per lane).
template <int NT, bool RELU, int NT0 = 0>
__device__ __forceinline__ void wg_store_t(const f32x4& acc, int ncol4, __bf16* lds_bf, int ld_bf, int nbf, float* lds_f, int ld_f,
                                           __bf16* __restrict__ hbm16, int ld_hbm, const int* row_r, int nc, int wave, int lane,
                                           unsigned long long* __restrict__ mb = nullptr, __bf16* lds_lo = nullptr) {
    const int nt = NT0 + wave;
    if (nt >= NT) return;                                        // wave-uniform
    const int col0 = nt * 16 + (lane >> 4) * 4, row = lane & 15;
    f32x4 v = acc;
    if (RELU) {
        if (mb) {
            const unsigned long long b0 = __ballot(acc[0] > 0.f), b1 = __ballot(acc[1] > 0.f), b2 = __ballot(acc[2] > 0.f), b3 = __ballot(acc[3] > 0.f);
            if (lane == 0) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2* q = reinterpret_cast<u64x2*>(mb + nt * 4);
                q[0] = (u64x2){b0, b1};
                q[1] = (u64x2){b2, b3};
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[r], 0.f);
    }
    if (col0 >= ncol4) return;
    if (lds_f) *reinterpret_cast<f32x4*>(lds_f + row * ld_f + col0) = v;
    if (lds_bf || hbm16) {
        const bf16x4 o = pack4(v[0], v[1], v[2], v[3]);
        if (lds_bf && col0 < nbf) *reinterpret_cast<bf16x4*>(lds_bf + row * ld_bf + col0) = o;
        if (lds_lo && col0 < nbf)      // the LOW part of the next split stage's operand: what the bf16 rounding of `o` dropped
            *reinterpret_cast<bf16x4*>(lds_lo + row * ld_bf + col0) = pack4(v[0] - (float)o[0], v[1] - (float)o[1], v[2] - (float)o[2], v[3] - (float)o[3]);
        if (hbm16 && row < nc) CH_GSTORE(reinterpret_cast<bf16x4*>(hbm16 + (size_t)row_r[row] * ld_hbm + col0), o);
    }
}

}  // namespace

// Everything a stage needs besides the streamed weights is in LDS before the stage starts: biases, the edge element, the
// cell tables and (IMG) the sample's image are loaded once per workgroup, the next wavefront's features and noise are
// fetched one wavefront ahead.  A global load inside a stage costs a full L2/HBM round trip on the critical path of a
// 46 x 19-stage dependent chain (measured ~0.5-1 us per stage) and, gfx9 counting loads and stores on one in-order
// counter, also drains the weight ring.
constexpr int BIAS_OFF[CW_COUNT] = {0, 112, 224, 336, 592, 720, 832, 944, 1056, 1168, 1280, 1392};
constexpr int BIAS_CNT[CW_COUNT] = {100, 100, NP + 8, 256, 128, 2 * A_, 100, 100, NP + 2, 100, 100, 1};
constexpr int BIAS_TOT = 1408;
constexpr int IMG_MAX = 128;              // side of the largest image staged in LDS (fp16)
constexpr int IMG_LD = IMG_MAX + 2;       // its row stride (65 dwords: consecutive rows start on different banks)

template <bool IMG>
__global__ __launch_bounds__(NTH) void k_chain_fwd(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float bias_sh[BIAS_TOT];
    __shared__ __attribute__((aligned(16))) float edge_sh[REC];
    // features of the next wavefront, parked at the end of the current one and read by its S0 stage only: in between the same floats stage
    // the rows' sd_attr (16 x 52) and stat (16 x 12) values for the store wave
    __shared__ __attribute__((aligned(16))) float feat_sh[MT][F];
    __shared__ __attribute__((aligned(16))) float noise_sh[MT][REC];      // [eps_box 4 | eps_attr A | eps_depth | logit(u_pres)]
    __shared__ unsigned short cell_hw[32 * 32];
    __shared__ float pbase_sh[32];                                         // base coordinate of glimpse index j (stn_base), no division per element
    __shared__ __attribute__((aligned(16))) unsigned long long mbf_sh[MB_TILES * 4];
    // fp16 copy of the sample's image with a ZERO guard column and row behind the last pixel (row stride IMG_LD): the second tap of a pair
    // that falls outside reads its zero from there -- no in-range test, select or mask multiply per tap, one address per element
    __shared__ __attribute__((aligned(16))) _Float16 img_sh[IMG ? (IMG_MAX + 1) * IMG_LD : 8];
    __shared__ __attribute__((aligned(16))) __bf16 Xc[MT * LD_XC];
    __shared__ __attribute__((aligned(16))) __bf16 XtZ[MT * LD_XT];
    __shared__ __attribute__((aligned(16))) __bf16 XtO[MT * LD_XT];
    __shared__ __attribute__((aligned(16))) __bf16 Gl[MT * LD_GL];
    __shared__ __attribute__((aligned(16))) __bf16 Ha[MT * LD_H];
    __shared__ __attribute__((aligned(16))) __bf16 Hb[MT * LD_H];
    __shared__ __attribute__((aligned(16))) float Ost[MT * LD_O];
    __shared__ __attribute__((aligned(16))) float recs[4][MT][REC];
    __shared__ __attribute__((aligned(16))) float nb_sh[MT][4];
    __shared__ __attribute__((aligned(16))) uint2 gtab[MT][2][PG];          // glimpse source coordinates per (row, axis, index)
    __shared__ int row_r[2][MT], row_hw[2][MT], row_cp[MT];                 // row tables of the current ([t & 1]) and the previous wavefront
    __shared__ int dstart_sh[3 * 32 + 2];
    __shared__ int bc0_sh[3 * 32 + 2], bnc_sh[3 * 32 + 2];                  // this band's first cell / cell count on every wavefront
    __shared__ __attribute__((aligned(16))) float brec_sh[4][REC];          // band split: the last records of the band above (ring over the grid column)
    __shared__ int ticket_sh;
    __shared__ short nbr_sh[32 * 32 * 4];
    __shared__ __attribute__((aligned(16))) float w2_sh[7 * 16];            // obj_network.out.weight as the bf16 values the MFMA path would multiply by (0 past column 99)
    __shared__ __attribute__((aligned(16))) float opart[MT][28];            // partial presence logits: row x (wave, column group of its tile) (OBJ1's epilogue)
    __shared__ float logit_sh[MT];

    const CellLayout& L = a.L;
    const CellBufs& P = a.P;
    const CellHyper& H = a.H;
    const int tid0 = threadIdx.x;
    int tid = tid0, lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int G = L.G, T = 3 * G - 2;
    // (sample, band) by START ORDER, not by blockIdx: the first B workgroups to start take the top bands, which wait for nobody; a workgroup
    // that has to wait therefore only ever waits for workgroups that are already running -- no assumption about dispatch order or residency
    const int NBn = a.nbands;
    int band = 0, b = blockIdx.x;
    if (NBn > 1) {
        if (tid0 == 0) ticket_sh = atomicAdd(&a.sync[0], 1);
        __syncthreads();
        const int tk = ticket_sh;
        band = tk / L.B;
        b = tk - band * L.B;
    }
    const int HBd = (G + NBn - 1) / NBn, hb0 = band * HBd, hb1 = min(G, hb0 + HBd);
    const int t_first = 2 * hb0, t_last = 2 * (hb1 - 1) + G - 1;
    int* const flag_mine = a.sync ? a.sync + CHAIN_SYNC_HDR + b * NBn + band : nullptr;
    // the LOW halves of the box network's split-bf16 operands live in the glimpse tile, which is idle until the box is known:
    // [feat | ctx] low (dead after BOX0; BOX1's output low part re-uses it) and BOX0's output low part behind it
    __bf16* const XcLo = Gl;
    __bf16* const HaLo = Gl + MT * LD_XC;
    __bf16* const HbLo = Gl;
    static_assert(MT * LD_XC + MT * LD_H <= MT * LD_GL, "low tiles fit the glimpse tile");
    float* const sd_stage = &feat_sh[0][0];                    // [MT][52]
    float* const stat_stage = &feat_sh[0][0] + MT * 52;        // [MT][12]: mu_box 4 | sd_box 4 | mu_depth | sd_depth | 0 | 0
    constexpr int LD_SD = 52, LD_ST = 12;
    static_assert(MT * (52 + 12) <= MT * F, "staging fits the feature tile");

    if (tid < 7 * 16) w2_sh[tid] = tid < 100 ? (float)(__bf16)a.w_obj2[tid] : 0.f;
    for (int i = tid; i < MT * LD_XC; i += NTH) Xc[i] = (__bf16)0.f;
    for (int i = tid; i < MT * LD_XT; i += NTH) { XtZ[i] = (__bf16)0.f; XtO[i] = (__bf16)0.f; }
    for (int i = tid; i < MT * LD_GL; i += NTH) Gl[i] = (__bf16)0.f;
    for (int i = tid; i < MT * LD_H; i += NTH) { Ha[i] = (__bf16)0.f; Hb[i] = (__bf16)0.f; }
    for (int i = tid; i < MT * LD_O; i += NTH) Ost[i] = 0.f;
    for (int i = tid; i <= T; i += NTH) {
        const int ds = P.diag_start[i];
        dstart_sh[i] = ds;
        // cells of wavefront i are ordered by grid row: rows hlo .. hhi, of which this band owns [max(hlo, hb0), min(hhi, hb1 - 1)]
        const int hlo = max(0, (i - G + 2) >> 1), hhi = min(G - 1, i >> 1);
        const int hA = max(hlo, hb0), hB = min(hhi, hb1 - 1);
        bc0_sh[i] = ds + (hA - hlo);
        bnc_sh[i] = i < T ? max(0, hB - hA + 1) : 0;
    }
    for (int i = tid; i < L.HW * 4; i += NTH) nbr_sh[i] = (short)P.nbr[i];
    for (int i = tid; i < L.HW; i += NTH) cell_hw[i] = (unsigned short)((P.cell_h[i] << 8) | P.cell_w[i]);
    for (int i = tid; i < REC; i += NTH) edge_sh[i] = P.edge[i];
    if (tid < PG) pbase_sh[tid] = stn_base(tid, PG, a.ac);
    // (every tile's accumulator starts from a whole bias quad: the padding between the layers' entries must read as zero)
#pragma unroll
    for (int l = 0; l < CW_COUNT; ++l) {
        const int span = (l + 1 < CW_COUNT ? BIAS_OFF[l + 1] : BIAS_TOT) - BIAS_OFF[l];
        for (int i = tid; i < span; i += NTH) bias_sh[BIAS_OFF[l] + i] = i < BIAS_CNT[l] ? a.bias[l][min(i, BIAS_CNT[l] - 1)] : 0.f;
    }
    if constexpr (IMG) {
        for (int i = tid; i < (a.I + 1) * IMG_LD / 2; i += NTH) reinterpret_cast<unsigned*>(img_sh)[i] = 0u;      // (the guards; IMG_LD is even)
        __syncthreads();
        const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)b * a.I * a.I);
        for (int i = tid; i < a.I * a.I / 4; i += NTH) {
            const float4 v = src[i];
            const float pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = i * 4 + k, y = p / a.I;
                img_sh[y * IMG_LD + (p - y * a.I)] = (_Float16)pv[k];
            }
        }
    }
    __syncthreads();

    // features and noise of wavefront tn, fetched into registers one wavefront ahead and parked in LDS at the end of the
    // current one: thread -> (row, 4 features) and 2 x (row, noise slot)
    float4 pf_feat = make_float4(0.f, 0.f, 0.f, 0.f);
    float pf_noise[2] = {0.f, 0.f};
    // per-thread constants of the prefetch (which feature quad / which noise slots this thread owns): computed ONCE -- the divisions and
    // the four-way source select were ~40 instructions of every wavefront's first stage
    const int pf_frow = min(tid0, MT * (F / 4) - 1) / (F / 4), pf_fc4 = (min(tid0, MT * (F / 4) - 1) - pf_frow * (F / 4)) * 4;
    int pf_nrow[2];
    const float* pf_nsrc[2];
    bool pf_isu[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = min(tid0 + q * NTH, MT * REC - 1);
        const int row = i / REC, j = i - row * REC;
        pf_nrow[q] = row;
        pf_isu[q] = j == REC - 1;
        pf_nsrc[q] = j < 4        ? P.eps_box + ((size_t)b * 4 + j) * G * G
                     : j < 4 + A_ ? P.eps_attr + ((size_t)b * A_ + (j - 4)) * G * G
                     : j == 4 + A_ ? P.eps_depth + (size_t)b * G * G
                                   : P.u_pres + (size_t)b * G * G;
    }
    auto prefetch = [&](int tn) {                 // branch-free (clamped indices): conditional loads would make every later wait on the
                                                  // weight ring a vmcnt(0), i.e. a wait for THESE loads
        const int c0n = bc0_sh[tn], ncn = bnc_sh[tn];
        {
            const int hw = cell_hw[c0n + min(pf_frow, ncn - 1)];
            pf_feat = *reinterpret_cast<const float4*>(P.feat + ((size_t)(b * G + (hw >> 8)) * G + (hw & 255)) * P.ld_feat + pf_fc4);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int hw = cell_hw[c0n + min(pf_nrow[q], ncn - 1)];
            pf_noise[q] = pf_nsrc[q][(size_t)(hw >> 8) * G + (hw & 255)];
        }
    };
    auto park = [&](int tidv) {
        if (tidv < MT * (F / 4)) {
            const int row = tidv / (F / 4), c4 = (tidv - row * (F / 4)) * 4;
            *reinterpret_cast<float4*>(&feat_sh[row][c4]) = pf_feat;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tidv + q * NTH;
            // the presence noise is parked as the logistic noise it becomes (models.py:402-409): log(u + 1e-9) - log(1 - u + 1e-9), here,
            // off the chain, instead of two logarithms inside the presence stage
            const float v = pf_isu[q] ? ch_log(pf_noise[q] + 1e-9f) - ch_log(1.f - pf_noise[q] + 1e-9f) : pf_noise[q];
            if (i < MT * REC) noise_sh[i / REC][i % REC] = v;
        }
    };
    prefetch(t_first);
    park(tid);

    // ---- band split: the hand-off of the band above's last row (one record per wavefront).  Producer: the store wave writes the record
    // with agent-scope (write-through) 8-byte stores, drains them, then publishes the wavefront count; consumer: one relaxed agent-scope
    // poll, then agent-scope 8-byte loads into registers (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table: one storing wave
    // signals for its own stores).  Every wait is bounded: on a time-out the error word is set and the kernel runs on (wrong, but it ends).
    typedef unsigned long long u64_t;
    int sync_dead = 0;
    auto publish_row = [&](int tp, int ncp, int ln) {            // called by the store wave behind flush_records(tp)
        if (NBn > 1 && band < NBn - 1) {
            const int w = tp - 2 * (hb1 - 1);                    // the band's last row on wavefront tp, if that wavefront reaches it
            if (w >= 0 && w < G && (tp >> 1) >= hb1 - 1) {
                if (ln < REC / 2) {
                    const u64_t v = *reinterpret_cast<const u64_t*>(&recs[tp & 3][ncp - 1][2 * ln]);
                    __hip_atomic_store(reinterpret_cast<u64_t*>(a.bnd_rec + ((size_t)(b * NBn + band) * G + w) * REC) + ln, v, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_store(flag_mine, tp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto fetch_row = [&](int tq, int ln) {                       // the band above's record of wavefront tq -> brec_sh (whole wave)
        const int w = tq - 2 * (hb0 - 1);
        if (w < 0 || w >= G) return;                             // (wave-uniform)
        const int* flag = a.sync + CHAIN_SYNC_HDR + b * NBn + (band - 1);
        int ok = sync_dead;
        for (int spin = 0; spin < CHAIN_SPIN_LIMIT && !ok; ++spin) {
            ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tq + 1);
            if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) { sync_dead = 1; if (ln == 0) { a.sync[2] = 1; a.sync[CHAIN_SYNC_STICKY(L.B, NBn)] = 1; } }      // time-out: never wait again, flag the launch (and, sticky, the workspace) as failed
        if (ln < REC / 2) {
            const u64_t v = __hip_atomic_load(reinterpret_cast<const u64_t*>(a.bnd_rec + ((size_t)(b * NBn + band - 1) * G + w) * REC) + ln,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<u64_t*>(&brec_sh[w & 3][2 * ln]) = v;
        }
    };
    if (band > 0 && wave == 7) {                                 // the two records the band's first wavefront reads
        fetch_row(t_first - 2, lane);
        fetch_row(t_first - 1, lane);
    }

    // The store wave's end-of-wavefront flush (one wavefront late, beside BOX0): the record rows, the box, the presence logit and the
    // two output maps of wavefront tp, whose row tables sit in slot tp & 1.  Everything a latent stage produces goes to LDS only.
    auto flush_records = [&](int tp, int ncp, int ln) {
        const int* rr = row_r[tp & 1];
        copy_rows_w<REC / 4, 16>(&recs[tp & 3][0][0], REC * 4, P.rec, (size_t)L.ld_rec * 4, rr, ncp, ln);
        if (ln < 4 * ncp) {        // lane (row, k): nbox element k and its z_where map entry
            const int row = ln >> 2, k = ln & 3, hw = row_hw[tp & 1][row];
            const float nv = nb_sh[row][k];
            P.nbox[(size_t)rr[row] * 4 + k] = nv;
            P.z_where[(((size_t)b * 4 + k) * G + (hw >> 8)) * G + (hw & 255)] = nv;
        }
        if (ln < ncp) {
            const int hw = row_hw[tp & 1][ln];
            P.Oo[(size_t)rr[ln] * L.ld_oo] = logit_sh[ln];
            P.z_pres[((size_t)b * G + (hw >> 8)) * G + (hw & 255)] = recs[tp & 3][ln][REC - 1];
        }
        publish_row(tp, ncp, ln);
    };

    WPipe pipe;
    // encoder layer 0's last ENC0_RES k-steps of this wave's two column tiles stay in registers for the whole kernel (see wg_gemm_t2)
    uint4 enc0_res[2 * ENC0_RES + 1];
    {
        const uint4* b0 = a.w[CW_ENC0] + (size_t)wave * 25 * 64 + lane;
        const uint4* b1 = a.w[CW_ENC0] + (size_t)(wave + 8) * 25 * 64 + lane;
#pragma unroll
        for (int i = 0; i < ENC0_RES; ++i) { enc0_res[2 * i] = b0[(size_t)(25 - ENC0_RES + i) * 64]; enc0_res[2 * i + 1] = b1[(size_t)(25 - ENC0_RES + i) * 64]; }
        enc0_res[2 * ENC0_RES] = make_uint4(0u, 0u, 0u, 0u);
    }
    pipe_fill_s<11, 7>(a.w[CW_BOX0], a.wlo[CW_BOX0], pipe, wave, lane);      // (all eight waves here: the ring registers must be defined on every path)
    int stamp_i = 0;
    int nc_prev = 0;
#define CH_STAMP() do { if (a.stamps && b == 0 && band == 0 && tid == 0) a.stamps[stamp_i++] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int t = t_first; t <= t_last; ++t) {
        // opaque per-iteration copies: keeps LICM from hoisting every layer's lane-dependent address arithmetic out of the
        // wavefront loop (that cost >100 VGPRs held across the whole loop and forced the weight ring to be shallow)
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        CH_STAMP();
        const int c0 = bc0_sh[t];
        const int nc = bnc_sh[t];
        unsigned long long* const mbt = mbf_sh;                 // sign-bit ballots of this wavefront, flushed to HBM at its end
        float (*rec_cur)[REC] = recs[t & 3];
        int* const rr_cur = row_r[t & 1];
        int* const hw_cur = row_hw[t & 1];
        if (tid < MT) {
            const int cp = c0 + min(tid, nc - 1);
            row_cp[tid] = cp;
            rr_cur[tid] = cp * L.B + b;
            hw_cur[tid] = cell_hw[cp];
        }
        lds_barrier();                       // also orders park() of the previous wavefront before S0's reads
        if constexpr (!IMG) prefetch(min(t + 1, t_last));      // (see the table stage)
        CH_STAMP();
        // ---- S0: [feat | context] (models.py:71-76,292-320), 4 floats per thread, LDS only (the store wave copies the finished rows to
        // the Xb row buffer beside BOX0)
        // (only the wavefront's nc live rows: the tile's other rows keep whatever finite values an earlier wavefront left -- their
        //  outputs are never stored)
        for (int idx = tid; idx < nc * ((F + CTX) / 4); idx += NTH) {
            const int row = idx / ((F + CTX) / 4), c4 = (idx - row * ((F + CTX) / 4)) * 4;
            float4 v;
            if (c4 < F) {
                v = *reinterpret_cast<const float4*>(&feat_sh[row][c4]);
            } else {
                const int s = (c4 - F) / REC, j = (c4 - F) - s * REC;
                const int nbc = nbr_sh[row_cp[row] * 4 + s];
                const int dt = (s == 0) ? 3 : (s == 1 ? 2 : 1);      // UL: t-3, U: t-2, UR and L: t-1
                const float* src = nbc >= 0 ? &recs[(t - dt) & 3][nbc - bc0_sh[max(t - dt, 0)]][j] : &edge_sh[j];
                if (band > 0 && nbc >= 0) {                          // a neighbour in the band above: its record came through brec_sh
                    const int nhw = cell_hw[nbc];
                    if ((nhw >> 8) < hb0) src = &brec_sh[nhw & 3][j];
                }
                v = *reinterpret_cast<const float4*>(src);
            }
            const bf16x4 hi = pack4(v.x, v.y, v.z, v.w);
            *reinterpret_cast<bf16x4*>(&Xc[row * LD_XC + c4]) = hi;
            *reinterpret_cast<bf16x4*>(&XcLo[row * LD_XC + c4]) = pack4(v.x - (float)hi[0], v.y - (float)hi[1], v.z - (float)hi[2], v.w - (float)hi[3]);
        }
        lds_barrier();
        CH_STAMP();
        // ---- z_where: box MLP (models.py:76-77)
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_s<11, 7, 4>(Xc, XcLo, LD_XC, a.w[CW_BOX0], a.wlo[CW_BOX0], pipe, bias_sh + BIAS_OFF[CW_BOX0], acc, wave, lane);
            pipe_fill_s<4, 7>(a.w[CW_BOX1], a.wlo[CW_BOX1], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Ha, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HB1 * 4, HaLo);
        } else {
            // (every operand of the weight-gradient GEMMs is stored as bf16 by this kernel: same leading dimensions in elements, the buffers
            //  are sized for the per-wavefront path's fp32; the z / obj nets read these columns from Xb too)
            if (t > t_first) flush_records(t - 1, nc_prev, lane);
        }
        lds_barrier();
        CH_STAMP();
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_s<4, 7, 4>(Ha, HaLo, LD_H, a.w[CW_BOX1], a.wlo[CW_BOX1], pipe, bias_sh + BIAS_OFF[CW_BOX1], acc, wave, lane);
            pipe_fill_s<4, 7>(a.w[CW_BOXH], a.wlo[CW_BOXH], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Hb, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HB2 * 4, HbLo);
        } else {
            copy_rows_w<25, 8>(Ha, LD_H * 2, P.Hb1, (size_t)SP_LDH * 2, rr_cur, nc, lane);
        }
        lds_barrier();
        CH_STAMP();
        // BOXH: [passthrough NP | 8 latents].  The passthrough columns go to the z-net's input tile (bf16) from the epilogue itself; the wave
        // that owns the latent columns (tile 6: columns 96..111) turns them into the box right behind its own stores -- no stage (barrier +
        // a 16-thread latent pass) of its own.
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_s<4, 7, 4>(Hb, HbLo, LD_H, a.w[CW_BOXH], a.wlo[CW_BOXH], pipe, bias_sh + BIAS_OFF[CW_BOXH], acc, wave, lane);
            wg_store_t<7, false>(acc, NP + 8, XtZ, LD_XT, NP, Ost, LD_O, nullptr, 0, rr_cur, nc, wave, lane);
        } else {
            copy_rows_w<25, 8>(Hb, LD_H * 2, P.Hb2, (size_t)SP_LDH * 2, rr_cur, nc, lane);
            copy_rows_w<(F + CTX + 4) / 8, 16>(Xc, LD_XC * 2, P.Xb, (size_t)L.ld_xb * 2, rr_cur, nc, lane);      // (beside wave 6's latent pass)
        }
        if constexpr (IMG) early_fill<16, 7, 0, 6>(a.w[CW_Z0], a.w[CW_OBJ0], pipe, wave, lane);      // early products, first six k-steps
        else pipe_fill2<25>(a.w[CW_ENC0], pipe, wave, lane);
        // ---- box latents (models.py:322-381): one latent per lane of wave 6; lane k of a row owns z_k -> (cell_y, cell_x, height, width)[k]
        // -> box / nbox element k ^ 1.  LDS only: records, stats and the box reach HBM through the store wave.
        if (wave == 6) {
            wave_lds_sync();                      // this wave's own Ost columns
            if (lane < 4 * nc) {
                const int row = lane >> 2, k = lane & 3, o = k ^ 1;
                const int hw = hw_cur[row];
                const float* lat = &Ost[row * LD_O + NP];
                const float mu = freeze_val(H.wheel, lat[k]);
                const float sd = freeze_val(H.wheel, 2.f * ch_sigmoid(clamp10(lat[4 + k])));
                const float sg = ch_sigmoid(clamp10(mu + sd * noise_sh[row][k]));
                float bv, nv;                               // box_forward (cell_math.h), element by element
                if (k < 2) {
                    bv = H.range_yx * sg + H.min_yx;                                    // cell_y (k = 0), cell_x (k = 1)
                    nv = H.cell_over_img * (bv + (float)(k == 0 ? (hw >> 8) : (hw & 255)));      // yt, xt
                } else {
                    bv = H.range_hw * sg + H.min_hw;                                    // height (k = 2), width (k = 3)
                    nv = bv * H.anchor / H.img;                                         // ys, xs (a true division, as the reference: the sampling
                                                                                        // grid's floor() decisions downstream are sensitive to the last bit)
                }
                stat_stage[row * LD_ST + ST_MU_BOX + k] = mu;
                stat_stage[row * LD_ST + ST_SD_BOX + k] = sd;
                rec_cur[row][o] = bv;
                nb_sh[row][o] = nv;
                XtZ[row * LD_XT + NP + o] = (__bf16)bv;
                XtO[row * LD_XT + NP + o] = (__bf16)bv;
            }
        }
        lds_barrier();
        CH_STAMP();
        // next wavefront's features and noise (parked at the end of this one): requested here, in front of the table stage, rather than at the
        // top of the wavefront in front of S0 (kernel 0.7055 -> 0.7020 ms; at the attribute stage 0.705, at the z_depth stage 0.710).  Not on images
        // sampled from global memory: the sampling's waits for its pixels would wait for these loads too (configs[3]: 1.607 -> 1.629 ms)
        if constexpr (IMG) prefetch(min(t + 1, t_last));
        // ---- z_what: glimpse (modules.py:216-273, border padding) + encoder MLP (models.py:383-391)
        // The source coordinates are separable: 28 column and 28 row coordinates per cell, tabulated first (one entry per thread:
        // first tap index | "second tap inside" | "not clipped", fractional weight) instead of being re-derived by every element
        // (5 coordinate evaluations per 4 elements were ~40 % of the sampling stage's instructions).  Waves 0..6 build the table; the
        // 8th copies the box head's output rows to HBM meanwhile.
        if (wave == 7) {
            copy_rows_w<(NP + 8) / 4, 16>(Ost, LD_O * 4, P.Ob, (size_t)L.ld_ob * 4, rr_cur, nc, lane);
        } else {
            for (int e = tid; e < nc * 2 * PG; e += 7 * 64) {
                const int row = e / (2 * PG), rem = e - row * (2 * PG), axis = rem >= PG ? 1 : 0, gi = rem - axis * PG;
                float cc, mm;
                stn_src_coord_b(nb_sh[row][axis ? 3 : 2], 2.f * nb_sh[row][axis ? 1 : 0] - 1.f, pbase_sh[gi], a.I, a.ac, true, cc, mm);
                const int c0 = (int)floorf(cc);
                gtab[row][axis][gi] = make_uint2((unsigned)c0 | ((c0 + 1) < a.I ? 0x10000u : 0u) | (mm != 0.f ? 0x20000u : 0u), __float_as_uint(cc - (float)c0));
            }
        }
        lds_barrier();
        const float gmult = a.ac ? 0.5f * (float)(a.I - 1) : 0.5f * (float)a.I;      // d(source pixel coordinate) / d(normalised coordinate) where not clipped
        // four glimpse elements (one row i, columns j0..j0+3 of one cell): bilinear taps, value + the two coordinate derivatives.
        // `buffered`: the HBM stores go through buffer descriptors with `live` as the lane mask (see the early products below)
        const __amdgpu_buffer_rsrc_t rs_gl = buf_rsrc(P.glimpse), rs_gxy = buf_rsrc(P.gxy);
        auto sample4 = [&](int idx, bool live, auto buffered) {
            const int row = idx / (GLN / 4), e = (idx - row * (GLN / 4)) * 4;
            const int i = e / PG, j0 = e - i * PG;              // P % 4 == 0: the 4 elements share the row i
            const uint2 ye = gtab[row][1][i];
            const int y0 = (int)(ye.x & 0xffffu);
            const float wy1 = __uint_as_float(ye.y), wy0 = 1.f - wy1;
            const bool yin = (ye.x & 0x10000u) != 0u;
            const float my = (ye.x & 0x20000u) ? gmult : 0.f;
            const int r0o = y0 * a.I, r1o = (yin ? y0 + 1 : y0) * a.I;
            const float* img = a.x + (size_t)b * a.I * a.I;
            const uint4 xa = *reinterpret_cast<const uint4*>(&gtab[row][0][j0]), xb = *reinterpret_cast<const uint4*>(&gtab[row][0][j0 + 2]);
            const unsigned xw[4] = {xa.x, xa.z, xb.x, xb.z}, xf[4] = {xa.y, xa.w, xb.y, xb.w};
            float out[4];
            unsigned int gxy[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x0 = (int)(xw[q] & 0xffffu);
                const bool xin = (xw[q] & 0x10000u) != 0u;
                const float mx = (xw[q] & 0x20000u) ? gmult : 0.f;
                const float wx1 = __uint_as_float(xf[q]), wx0 = 1.f - wx1;
                float v00, v01, v10, v11;
                if constexpr (IMG) {
                    // (a tap outside reads the zero guard: the same value as the masked tap.  volatile: left alone the compiler merges the tap
                    //  pairs into 4-byte reads at 2-byte-aligned addresses -- legal on this target and much slower: sampling stage 3.9 -> 6.2 us)
                    typedef const volatile __attribute__((address_space(3))) _Float16* lds_tap_t;
                    const lds_tap_t t0 = (lds_tap_t)(img_sh + (y0 * IMG_LD + x0));
                    v00 = (float)t0[0]; v01 = (float)t0[1]; v10 = (float)t0[IMG_LD]; v11 = (float)t0[IMG_LD + 1];
                } else {
                    // the two taps of a row as ONE 8-byte load (4-byte aligned: global memory takes it): half the loads of the stage that the
                    // vector-memory front end bounds on images too wide for the LDS copy.  At the right border (x0 = I - 1, second tap masked)
                    // the pair starts one pixel to the left.
                    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
                    const float m1 = xin ? 1.f : 0.f, n1 = yin ? 1.f : 0.f;
                    const int xl = min(x0, a.I - 2);
                    const f2u p0 = *reinterpret_cast<const f2u*>(img + r0o + xl), p1 = *reinterpret_cast<const f2u*>(img + r1o + xl);
                    const bool sh = x0 != xl;
                    v00 = sh ? p0.y : p0.x; v01 = m1 * p0.y;
                    v10 = n1 * (sh ? p1.y : p1.x); v11 = m1 * n1 * p1.y;
                }
                // nested interpolation: the row differences d0, d1 and the column difference of the two interpolated rows are also the
                // derivatives d val / d (normalised source x, y) -- what the backward pass needs instead of re-gathering the image --
                // (10 operations for value + both derivatives; 18 as four weighted taps + two difference sums)
                const float d0 = v01 - v00, d1 = v11 - v10;
                const float top = fmaf(wx1, d0, v00), bot = fmaf(wx1, d1, v10), db = bot - top;
                out[q] = fmaf(wy1, db, top);
                const float gx = fmaf(wy1, d1 - d0, d0) * mx;
                const float gy = db * my;
                union { _Float16 h[2]; unsigned int u; } pk;      // fp16 pair: |g| <= I/2 fits, 11 significant bits
                pk.h[0] = (_Float16)gx; pk.h[1] = (_Float16)gy;
                gxy[q] = pk.u;
            }
            bf16x4 o;
            o[0] = (__bf16)out[0]; o[1] = (__bf16)out[1]; o[2] = (__bf16)out[2]; o[3] = (__bf16)out[3];
            *reinterpret_cast<bf16x4*>(&Gl[row * LD_GL + e]) = o;      // (a masked lane re-computes the stage's last element: same value, same address)
            if constexpr (decltype(buffered)::value) {
                const unsigned eo = (unsigned)rr_cur[row] * (unsigned)L.ld_gl + (unsigned)e;      // every row buffer is < 4 GB
                const u32x2_t ob = __builtin_bit_cast(u32x2_t, o);
                buf_store8_nt(rs_gl, live ? eo * 2u : BUF_OOB, ob.x, ob.y);
                buf_store16_nt(rs_gxy, live ? eo * 4u : BUF_OOB, (u32x4_t){gxy[0], gxy[1], gxy[2], gxy[3]});
            } else {
                CH_GSTORE(reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.glimpse) + (size_t)rr_cur[row] * L.ld_gl + e), o);
                CH_GSTORE(reinterpret_cast<u32x4_t*>(P.gxy + (size_t)rr_cur[row] * L.ld_gl + e), ((u32x4_t){gxy[0], gxy[1], gxy[2], gxy[3]}));
            }
        };
        f32x4 accZ = {0.f, 0.f, 0.f, 0.f}, accO = {0.f, 0.f, 0.f, 0.f};
        if constexpr (IMG) {
            // The sampling rounds with the early products of Z0 / OBJ0 between them (see wg_gemm_tk).  A round is wave-uniform code without
            // lane predication: lanes past the stage's last element re-compute it and mask their HBM stores through the descriptor, a round
            // past the end issues the two stores fully masked -- every path carries the same number of memory operations, so the waits
            // for the weight fragments stay counted (vmcnt(n)) instead of collapsing to vmcnt(0) at each join, which would also wait for
            // the acknowledgement of the round's own stores.
            const int lim = __builtin_amdgcn_readfirstlane(nc) * (GLN / 4);
            auto round = [&](int it) {
                if (it * NTH + wave * 64 < lim) sample4(min(tid + it * NTH, lim - 1), tid + it * NTH < lim, std::true_type{});      // (per wave)
                else {      // (masked: the data operand is whatever is live)
                    const u32x4_t junk = __builtin_bit_cast(u32x4_t, accZ);
                    buf_store8_nt(rs_gl, BUF_OOB, junk.x, junk.y);
                    buf_store16_nt(rs_gxy, BUF_OOB, junk);
                }
            };
            const int ntz = min(wave, 6);
            const __bf16* pxc = Xc + (lane & 15) * LD_XC + (lane >> 4) * 8;
            accZ = *reinterpret_cast<const f32x4*>(bias_sh + BIAS_OFF[CW_Z0] + ntz * 16 + (lane >> 4) * 4);
            accO = *reinterpret_cast<const f32x4*>(bias_sh + BIAS_OFF[CW_OBJ0] + ntz * 16 + (lane >> 4) * 4);
            // (the activation fragments are read behind each round, not ahead of it: 24 registers live across a sampling round spilled, and
            //  a spill reload inside the round waits -- in-order vmcnt -- for every weight fragment issued before it)
            round(0);
            {
                bf16x8 xe[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) xe[j] = *reinterpret_cast<const bf16x8*>(pxc + j * 32);
                early_mfma<6>(xe, pipe, accZ, accO);
            }
            early_fill<16, 7, 6, 5>(a.w[CW_Z0], a.w[CW_OBJ0], pipe, wave, lane);
            round(1);
            round(2);
            {
                bf16x8 xe[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) xe[j] = *reinterpret_cast<const bf16x8*>(pxc + (6 + j) * 32);
                early_mfma<5>(xe, pipe, accZ, accO);
            }
            pipe_fill2<25>(a.w[CW_ENC0], pipe, wave, lane);
            round(3);
            for (int idx = tid + 4 * NTH; idx < lim; idx += NTH) sample4(idx, true, std::false_type{});      // (more than 10 cells per wavefront: not with G <= 16)
        } else {
            for (int idx = tid; idx < nc * (GLN / 4); idx += NTH) sample4(idx, true, std::false_type{});
        }
        lds_barrier();
        CH_STAMP();
        {   // 256 outputs = 16 tiles, two per wave sharing every glimpse fragment
            f32x4 acc0, acc1;
            wg_gemm_t2<25, 5, ENC0_RES>(Gl, LD_GL, a.w[CW_ENC0], pipe, enc0_res, bias_sh + BIAS_OFF[CW_ENC0], acc0, acc1, wave, lane);
            pipe_fill<8, 8>(a.w[CW_ENC1], pipe, wave, lane);
            wg_store_t<16, true, 0>(acc0, 256, Ha, LD_H, 256, nullptr, 0, reinterpret_cast<__bf16*>(P.He1), SP_ENC_H1, rr_cur, nc, wave, lane, mbt + MB_HE1 * 4);
            wg_store_t<16, true, 8>(acc1, 256, Ha, LD_H, 256, nullptr, 0, reinterpret_cast<__bf16*>(P.He1), SP_ENC_H1, rr_cur, nc, wave, lane, mbt + MB_HE1 * 4);
        }
        lds_barrier();
        CH_STAMP();
        {
            f32x4 acc;
            wg_gemm_t<8, 0, 8>(Ha, LD_H, nullptr, 0, a.w[CW_ENC1], pipe, bias_sh + BIAS_OFF[CW_ENC1], acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_ENC2], pipe, wave, lane);
            wg_store_t<8, true>(acc, 128, Hb, LD_H, 128, nullptr, 0, reinterpret_cast<__bf16*>(P.He2), SP_ENC_H2, rr_cur, nc, wave, lane, mbt + MB_HE2 * 4);
        }
        lds_barrier();
        CH_STAMP();
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_t<4, 0, 7>(Hb, LD_H, nullptr, 0, a.w[CW_ENC2], pipe, bias_sh + BIAS_OFF[CW_ENC2], acc, wave, lane);
            if constexpr (IMG) pipe_fill_k<16, 11, 7>(a.w[CW_Z0], pipe, wave, lane);
            else pipe_fill<16, 7>(a.w[CW_Z0], pipe, wave, lane);
            wg_store_t<7, false>(acc, 2 * A_, nullptr, 0, 0, Ost, LD_O, nullptr, 0, rr_cur, nc, wave, lane);
        }
        lds_barrier();
        CH_STAMP();
        // ---- attributes (models.py:83-85): LDS only
        if (wave == 7) copy_rows_w<(2 * A_) / 4, 16>(Ost, LD_O * 4, P.Oe, (size_t)L.ld_oe * 4, rr_cur, nc, lane);
        for (int idx = tid; idx < nc * A_; idx += NTH) {
            const int row = idx / A_, j = idx - row * A_;
            const float sd = 2.f * ch_sigmoid(clamp10(Ost[row * LD_O + A_ + j]));
            const float attr = Ost[row * LD_O + j] + sd * noise_sh[row][4 + j];
            const __bf16 ab = (__bf16)attr;
            rec_cur[row][4 + j] = attr;
            XtZ[row * LD_XT + NP + 4 + j] = ab;
            XtO[row * LD_XT + NP + 4 + j] = ab;
            sd_stage[row * LD_SD + j] = sd;
            if (j < 2) sd_stage[row * LD_SD + A_ + j] = 0.f;      // the copied row is 52 wide
        }
        lds_barrier();
        CH_STAMP();
        // ---- z_depth (models.py:88-97); the store wave: the z-net's whole input tail [pass | box | attr] (a stage later: the decoder's input
        // rows -- the attr columns of the same tile, their two pad columns being the tile's zero depth / pad slots -- and sd_attr)
        if (wave < 7) {
            f32x4 acc;
            if constexpr (IMG) { acc = accZ; wg_gemm_tk<16, 11>(XtZ, LD_XT, pipe, acc, lane); }
            else wg_gemm_t<11, 5, 7>(Xc, LD_XC, XtZ, LD_XT, a.w[CW_Z0], pipe, bias_sh + BIAS_OFF[CW_Z0], acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_Z1], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Ha, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HZ1 * 4);
        } else {
            copy_rows_w<39, 8>(XtZ, LD_XT * 2, reinterpret_cast<__bf16*>(P.Xz) + L.x_pass, (size_t)L.ld_x * 2, rr_cur, nc, lane);
        }
        lds_barrier();
        CH_STAMP();
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_t<4, 0, 7>(Ha, LD_H, nullptr, 0, a.w[CW_Z1], pipe, bias_sh + BIAS_OFF[CW_Z1], acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_ZH], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Hb, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HZ2 * 4);
        } else {
            copy_rows_w<25, 8>(Ha, LD_H * 2, P.Hz1, (size_t)SP_LDH * 2, rr_cur, nc, lane);
            copy_rows_w<13, 8>(XtZ + NP + 4, LD_XT * 2, P.Za16, (size_t)L.ld_rec * 2, rr_cur, nc, lane);
            copy_rows_w<13, 16>(sd_stage, LD_SD * 4, P.sd_attr, (size_t)L.ld_rec * 4, rr_cur, nc, lane);
        }
        lds_barrier();
        CH_STAMP();
        // ZH: [passthrough NP | depth mean, log-std]: passthrough -> obj-net input tile from the epilogue, depth by the wave that owns tile 6
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_t<4, 0, 7>(Hb, LD_H, nullptr, 0, a.w[CW_ZH], pipe, bias_sh + BIAS_OFF[CW_ZH], acc, wave, lane);
            if constexpr (IMG) pipe_fill_k<16, 11, 7>(a.w[CW_OBJ0], pipe, wave, lane);
            else pipe_fill<16, 7>(a.w[CW_OBJ0], pipe, wave, lane);
            wg_store_t<7, false>(acc, NP + 4, XtO, LD_XT, NP, Ost, LD_O, nullptr, 0, rr_cur, nc, wave, lane);
        } else {
            copy_rows_w<25, 8>(Hb, LD_H * 2, P.Hz2, (size_t)SP_LDH * 2, rr_cur, nc, lane);
        }
        if (wave == 6) {
            wave_lds_sync();
            if (lane < nc) {
                const float mu = freeze_val(H.wheel, Ost[lane * LD_O + NP]);
                const float sd = freeze_val(H.wheel, 2.f * ch_sigmoid(clamp10(Ost[lane * LD_O + NP + 1])));
                const float depth = 4.f * ch_sigmoid(clamp10(mu + sd * noise_sh[lane][4 + A_]));
                stat_stage[lane * LD_ST + ST_MU_DEPTH] = mu;
                stat_stage[lane * LD_ST + ST_SD_DEPTH] = sd;
                stat_stage[lane * LD_ST + 10] = 0.f;
                stat_stage[lane * LD_ST + 11] = 0.f;
                rec_cur[lane][4 + A_] = depth;
                XtO[lane * LD_XT + NP + 4 + A_] = (__bf16)depth;
            }
        }
        lds_barrier();
        CH_STAMP();
        // ---- z_pres (models.py:100-102,393-411); the store wave: z head rows, the obj-net's input tail [pass | box | attr | depth]
        if (wave < 7) {
            f32x4 acc;
            if constexpr (IMG) { acc = accO; wg_gemm_tk<16, 11>(XtO, LD_XT, pipe, acc, lane); }
            else wg_gemm_t<11, 5, 7>(Xc, LD_XC, XtO, LD_XT, a.w[CW_OBJ0], pipe, bias_sh + BIAS_OFF[CW_OBJ0], acc, wave, lane);
            pipe_fill<4, 7>(a.w[CW_OBJ1], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Ha, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HO1 * 4);
        } else {
            copy_rows_w<(NP + 4) / 4, 16>(Ost, LD_O * 4, P.Oz, (size_t)L.ld_oz * 4, rr_cur, nc, lane);      // 102 used columns, the row holds 104
        }
        lds_barrier();
        CH_STAMP();
        // this row's presence noise, read now: the pres stage below shares its barrier interval with park(), which overwrites noise_sh
        const float u_pres_reg = noise_sh[min(tid, MT - 1)][4 + A_ + 1];
        // OBJ1, and the one-column output layer with it: each lane multiplies its four (bf16-rounded, as the MFMA would see them) hidden values by
        // the output weight of its column -- a 100 -> 1 layer as an MFMA stage of its own cost a full barrier interval (0.9 us) for one
        // useful output column.  One partial per (row, wave, column group); the presence stage adds the 28 of a row.
        if (wave < 7) {
            f32x4 acc;
            wg_gemm_t<4, 0, 7>(Ha, LD_H, nullptr, 0, a.w[CW_OBJ1], pipe, bias_sh + BIAS_OFF[CW_OBJ1], acc, wave, lane);
            pipe_fill_s<11, 7>(a.w[CW_BOX0], a.wlo[CW_BOX0], pipe, wave, lane);
            wg_store_t<7, true>(acc, 100, Hb, LD_H, 100, nullptr, 0, nullptr, 0, rr_cur, nc, wave, lane, mbt + MB_HO2 * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(w2_sh + wave * 16 + (lane >> 4) * 4);
            float pz = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) pz = fmaf((float)(__bf16)fmaxf(acc[r], 0.f), w2[r], pz);
            opart[lane & 15][wave * 4 + (lane >> 4)] = pz;
        } else {
            copy_rows_w<25, 8>(Ha, LD_H * 2, P.Ho1, (size_t)SP_LDH * 2, rr_cur, nc, lane);
            copy_rows_w<3, 16>(stat_stage, LD_ST * 4, P.stat, (size_t)SP_LDSTAT * 4, rr_cur, nc, lane);
            copy_rows_w<39, 8>(XtO, LD_XT * 2, reinterpret_cast<__bf16*>(P.Xo) + L.x_pass, (size_t)L.ld_x * 2, rr_cur, nc, lane);
            if (band > 0) fetch_row(t, lane);      // the band above's record of this wavefront: read by the next wavefront's S0
        }
        lds_barrier();
        CH_STAMP();
        if (tid < nc) {
            float logit = bias_sh[BIAS_OFF[CW_OBJ2]];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const f32x4 pq = *reinterpret_cast<const f32x4*>(&opart[tid][q * 4]);
                logit += (pq[0] + pq[1]) + (pq[2] + pq[3]);
            }
            // pres_forward (cell_math.h) with the logistic noise already formed at park time
            const float pres = ch_sigmoid(clamp10(freeze_val(H.wheel, logit)) + u_pres_reg);
            logit_sh[tid] = logit;
            rec_cur[tid][REC - 1] = pres;
        }
        if (wave == 7) copy_rows_w<25, 8>(Hb, LD_H * 2, P.Ho2, (size_t)SP_LDH * 2, rr_cur, nc, lane);
        // (no barrier: nothing below reads what the pres threads write, and they no longer read noise_sh; the next wavefront's row-setup
        //  barrier orders all of it before S0)
        park(tid);                           // features / noise of the next wavefront (read after its row-setup barrier)
        if (tid < MB_TILES * 4) P.mbits[((size_t)(b * NBn + band) * T + t) * (MB_TILES * 4) + tid] = mbf_sh[tid];     // one coalesced 2 KB store
        nc_prev = nc;
        CH_STAMP();
    }
    lds_barrier();
    if (wave == 7) flush_records(t_last, nc_prev, tid0 & 63);
}


// =============================================================================================
// Backward: the same ownership (one workgroup per sample), wavefronts in reverse order.  Every layer-output gradient is written
// to its row buffer as BF16 (same leading dimension in elements; the buffers are sized for fp32, the per-wavefront path keeps using
// them as fp32): the weight-gradient GEMMs round their operands to bf16 anyway, and one grouped bf16-A launch replaces 14.  Data-gradient
// chain only -- every layer's pre-activation gradient is written to its HBM row buffer and the weight
// gradients are long-K GEMMs over all rows afterwards.  The input gradients of the three per-cell nets
// never go to HBM: their [feat | context] part is summed in a 4-deep LDS ring (a cell's record gradient
// is gathered from its <= 4 consumers' context columns, which live in wavefronts t+1..t+3), the feature
// part is emitted as d feat, out-of-grid context slots accumulate the edge element's gradient in LDS.
// The glimpse gradient is consumed in the epilogue of the encoder's first-layer data-gradient GEMM
// (bilinear taps of x -> d z_where) and is never stored.
// =============================================================================================
namespace {

constexpr int LD_R = 328;      // ring row: [feat 100 | ctx 224] fp32

// ---- backward GEMM stages, lean transposed form (see wg_gemm_t): out^T[n][row] = sum_k Wt[n][k] * in[row][k] with a wide N (7 .. 49 column
// tiles, tile nt = wave + 8 j for this wave's j-th tile) and a short K (4 or 8 k-steps).  The gradient tile's fragments are read once per
// stage; the weight fragments of ALL of this wave's tiles run through the same RD-deep register ring as in the forward kernel, in
// (tile, k-step) order, refilled behind each MFMA; `mid()` is called once behind the stage's last MFMA and before its last epilogue -- the
// place where the NEXT layer's first ring fill is issued (round 3 issued it behind the epilogue, at the barrier: every small stage then began
// by waiting out an L2 round trip).  `epi(j, nt, acc)`: lane holds columns nt*16 + (lane>>4)*4 .. +3 of row lane & 15.
// The backward kernel's ring is RDB = 8 deep.  Measured at 4 / 6 / 8 / 10 / 12 (same box, tools/exp/chain_ablate.py): 0.80 / 0.761 / 0.746 /
// 0.757 / 0.775 ms -- at 12 the kernel holds 246 of 256 registers and its one-tile-per-wave stages run 0.9 us instead of 0.6-0.7; below 8 the
// wide stages (encoder layer 0, the 30-tile first layers) starve.  Per-layer depths (6 for the narrow, 12 for the wide stages) were tried:
// the register count, not the depth, is what the narrow stages feel (17.6 us per wavefront against 17.0).
constexpr int RDB = 8;
struct WPipeB { uint4 q[RDB]; };
// A wave only streams (and multiplies) the tiles it owns: with NT not a multiple of the 8 waves the last round's surplus waves used to run a
// clamped duplicate of tile NT - 1 whose result was dropped -- 7 of encoder layer 0's 56 tile slots, one of 8 in every 7-tile layer: 104 KiB
// of the ~1.1 MB a wavefront streams.  The test is wave-uniform (a scalar branch).
template <int KT, int NT>
__device__ __forceinline__ void pipe_fill_w(const uint4* __restrict__ Wt, WPipeB& p, int wave, int lane) {
    constexpr int MY = (NT + NW - 1) / NW, TOT = MY * KT;
#pragma unroll
    for (int s_ = 0; s_ < RDB; ++s_)
        if (s_ < TOT && wave + NW * (s_ / KT) < NT) p.q[s_] = Wt[((size_t)(wave + NW * (s_ / KT)) * KT + (s_ % KT)) * 64 + lane];
}
template <int KT, int NT, class Epi, class Mid>
__device__ __forceinline__ void wg_gemm_wt(const __bf16* in, int ld, const uint4* __restrict__ Wt, WPipeB& p, int wave, int lane, Epi epi, Mid mid) {
    constexpr int MY = (NT + NW - 1) / NW, TOT = MY * KT;
    // every wave owns its tiles of the first MY - 1 rounds; the last round's tile only if it exists (wave-uniform: one scalar branch)
    const bool last_own = (NT % NW) == 0 || wave + NW * (MY - 1) < NT;
    const __bf16* pa = in + (lane & 15) * ld + (lane >> 4) * 8;
    bf16x8 x[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) x[kt] = *reinterpret_cast<const bf16x8*>(pa + kt * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < MY; ++j) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool own = j < MY - 1 || last_own;
        if (own) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int s_ = j * KT + kt;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(p.q[s_ % RDB]), x[kt], acc, 0, 0, 0);
                if (s_ + RDB < TOT && ((s_ + RDB) / KT < MY - 1 || last_own))
                    p.q[s_ % RDB] = Wt[((size_t)(wave + NW * ((s_ + RDB) / KT)) * KT + ((s_ + RDB) % KT)) * 64 + lane];
            }
        }
        // mid(): once, behind this wave's last MFMA
        if (MY >= 2 && j == MY - 2 && !last_own) mid();
        if (j == MY - 1 && (last_own || MY == 1)) mid();
        if (own) epi(j, wave + NW * j, acc);
    }
}

// One hidden layer of the data-gradient chain: dPre[16, nout] = (dOut . W) * relu'(H).  The relu masks are the sign bits the forward
// kernel left as wave ballots (mb[tile*4 + r], bit = lane): the MFMA output layout is the same in both kernels, so a lane tests its
// own bit of the tile's four words.  All 66 tiles x 4 words of a wavefront (2 KB) are fetched with one 8-byte load per thread and parked in LDS.
static_assert(NTH >= MB_TILES * 4, "one sign-bit word per thread");
// backward bundle row (floats): encoder output [mean A | logstd A] (104) | sd_attr (56) | g_attr from the decoder (56) | eps_attr (56)
// | nbox 4 | g_nbox 4 | box logstd 4 | stat 12 | eps_box 4 | eps_depth, z_pres, g_pres, obj logit, g_depth, depth mean, depth logstd
constexpr int BD_OE = 0, BD_SD = 104, BD_GA = 160, BD_EA = 216, BD_NB = 272, BD_GNB = 276, BD_OBL = 280, BD_ST = 284, BD_EB = 296,
              BD_EPSD = 300, BD_ZP = 301, BD_GPR = 302, BD_OO = 303, BD_GDR = 304, BD_OZ0 = 305, BD_OZ1 = 306, BD_W = 308;

template <int NT>
__device__ __forceinline__ void hidden_epi(int nt, const f32x4& acc, const unsigned long long* mb, int ncol4, __bf16* dst, __bf16* __restrict__ dOut16,
                                           int ldh, const int* row_r, int nc, int lane) {
    if (nt >= NT) return;                                        // wave-uniform
    const int col0 = nt * 16 + (lane >> 4) * 4, row = lane & 15;
    if (col0 >= ncol4) return;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 m01 = *reinterpret_cast<const u64x2*>(mb + nt * 4), m23 = *reinterpret_cast<const u64x2*>(mb + nt * 4 + 2);
    const float v0 = ((m01[0] >> lane) & 1ull) ? acc[0] : 0.f, v1 = ((m01[1] >> lane) & 1ull) ? acc[1] : 0.f;
    const float v2 = ((m23[0] >> lane) & 1ull) ? acc[2] : 0.f, v3 = ((m23[1] >> lane) & 1ull) ? acc[3] : 0.f;
    const bf16x4 o = pack4(v0, v1, v2, v3);
    *reinterpret_cast<bf16x4*>(dst + row * LD_H + col0) = o;
    if (dOut16 && row < nc) *reinterpret_cast<bf16x4*>(dOut16 + (size_t)row_r[row] * ldh + col0) = o;
}

// chain-local fast forms of the latent gradients (cell_math.h holds the reference forms; see ch_sigmoid)
__device__ __forceinline__ float ch_kl_gauss(float mu, float sd, float m, float rs) {      // rs = 1 / prior std
    const float q = sd * rs, vr = q * q, d = (mu - m) * rs;
    return 0.5f * (vr + d * d - 1.f - ch_log(vr));
}

}  // namespace

__global__ __launch_bounds__(NTH) void k_chain_bwd(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float ring[4][MT][LD_R];
    __shared__ __attribute__((aligned(16))) float tailO[MT][KX];
    __shared__ __attribute__((aligned(16))) float tailZ[MT][KX];
    __shared__ __attribute__((aligned(16))) __bf16 Aa[MT * LD_H];
    __shared__ __attribute__((aligned(16))) __bf16 Ab[MT * LD_H];
    __shared__ float grec[MT][REC];
    __shared__ __attribute__((aligned(16))) float gnbw[NW][MT][4];      // per-WAVE partial d nbox of the glimpse epilogue, summed in wave order (no atomics: run-to-run identical)
    __shared__ __attribute__((aligned(16))) float nb_sh[MT][4];
    __shared__ float zp_sh[MT];
    __shared__ __attribute__((aligned(16))) unsigned long long mb_sh[MB_TILES * 4];
    __shared__ int row_r[MT], row_h[MT], row_w[MT];
    __shared__ __attribute__((aligned(16))) int cons_sh[MT][4];
    __shared__ int nbr_row[MT][4];
    // per-row scalars and vectors of the wavefront, fetched one wavefront AHEAD with coalesced loads and parked here: no global
    // load is left inside a stage (each one cost a full HBM round trip on the critical path: 1.5-3 us in the pres / depth / attr /
    // box stages).  Row layout (floats): see BD_* below.
    __shared__ __attribute__((aligned(16))) float bundle_sh[MT][BD_W];
    __shared__ __attribute__((aligned(16))) int ibundle_sh[MT][8];           // consumers[4] | neighbours[4]
    __shared__ int dstart_sh[3 * 32 + 2];
    __shared__ int bc0_sh[3 * 32 + 2], bnc_sh[3 * 32 + 2];                  // this band's first cell / cell count on every wavefront
    __shared__ __attribute__((aligned(16))) float bgrad_sh[3][REC];         // band split: context-gradient pieces of the band below for this band's last row
    __shared__ int ticket_sh;
    __shared__ __attribute__((aligned(16))) float wobj_sh[SP_H + 12];
    __shared__ float pbase_sh[32];          // base coordinate of glimpse index j (stn_base)
    __shared__ float prior_sh[18];          // prior mean[6] | std[6] | 1 / std[6]: a lane-indexed read of the kernel-argument struct would be a
                                            // global load + vmcnt(0) wait in the middle of the prefetch window

    const CellLayout& L = a.L;
    const CellBufs& P = a.P;
    const CellHyper& H = a.H;
    const int tid0 = threadIdx.x;
    int tid = tid0, lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int G = L.G, T = 3 * G - 2;
    // (sample, band) by start order, the BOTTOM band first (see k_chain_fwd): the backward pipeline runs upwards
    const int NBn = a.nbands;
    int band = 0, b = blockIdx.x;
    if (NBn > 1) {
        if (tid0 == 0) ticket_sh = atomicAdd(&a.sync[1], 1);
        __syncthreads();
        const int tk = ticket_sh;
        band = NBn - 1 - tk / L.B;
        b = tk - (tk / L.B) * L.B;
    }
    const int HBd = (G + NBn - 1) / NBn, hb0 = band * HBd, hb1 = min(G, hb0 + HBd);
    const int t_first = 2 * hb0, t_last = 2 * (hb1 - 1) + G - 1;
    int* const flag_mine = a.sync ? a.sync + CHAIN_SYNC_HDR + L.B * NBn + b * NBn + band : nullptr;
    const float ks = H.kl_scale * (*P.gloss);
    if (tid < SP_H) wobj_sh[tid] = a.w_obj2[tid];
    if (tid < PG) pbase_sh[tid] = stn_base(tid, PG, a.ac);
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { prior_sh[i] = H.prior_mean[i]; prior_sh[6 + i] = H.prior_std[i]; prior_sh[12 + i] = 1.f / H.prior_std[i]; }
    }

    for (int i = tid; i < MT * LD_H; i += NTH) { Aa[i] = (__bf16)0.f; Ab[i] = (__bf16)0.f; }
    for (int i = tid; i < 4 * MT * LD_R; i += NTH) (&ring[0][0][0])[i] = 0.f;
    for (int i = tid; i < MT * KX; i += NTH) { (&tailO[0][0])[i] = 0.f; (&tailZ[0][0])[i] = 0.f; }
    for (int i = tid; i <= T; i += NTH) {
        const int ds = P.diag_start[i];
        dstart_sh[i] = ds;
        const int hlo = max(0, (i - G + 2) >> 1), hhi = min(G - 1, i >> 1);
        const int hA = max(hlo, hb0), hB = min(hhi, hb1 - 1);
        bc0_sh[i] = ds + (hA - hlo);
        bnc_sh[i] = i < T ? max(0, hB - hA + 1) : 0;
    }
    float edge_reg = 0.f;                    // threads NTH/2 .. NTH/2 + 4*REC: running sum of one (neighbour slot, record element) of the edge gradient
    __syncthreads();

    // ---- bundle prefetch: thread (row = tid>>5, l = tid&31) owns two 16-byte and two 4-byte items of that row.  Bases, strides
    // and LDS slots are fixed per thread (computed once here); only the row index changes per wavefront.
    const int brow = tid0 >> 5, bl = tid0 & 31;
    const float *v0_base, *v1_base, *s1_base;
    int v0_ld, v0_dst, v1_ld, v1_dst, v1_cp = 0, s1_stride, s1_dst, s1_row;
    if (bl < 26) { v0_base = P.Oe + bl * 4; v0_ld = L.ld_oe; v0_dst = BD_OE + bl * 4; }
    else { v0_base = P.sd_attr + (bl - 26) * 4; v0_ld = L.ld_rec; v0_dst = BD_SD + (bl - 26) * 4; }
    if (bl < 8) { v1_base = P.sd_attr + (6 + bl) * 4; v1_ld = L.ld_rec; v1_dst = BD_SD + (6 + bl) * 4; }
    else if (bl < 22) { v1_base = P.g_attr_r + (bl - 8) * 4; v1_ld = L.ld_rec; v1_dst = BD_GA + (bl - 8) * 4; }
    else if (bl == 22) { v1_base = P.nbox; v1_ld = 4; v1_dst = BD_NB; }
    else if (bl == 23) { v1_base = P.g_nbox_r; v1_ld = 4; v1_dst = BD_GNB; }
    else if (bl == 24) { v1_base = P.Ob + NP + 4; v1_ld = L.ld_ob; v1_dst = BD_OBL; }
    else if (bl < 28) { v1_base = P.stat + (bl - 25) * 4; v1_ld = SP_LDSTAT; v1_dst = BD_ST + (bl - 25) * 4; }
    else if (bl == 28) { v1_base = reinterpret_cast<const float*>(P.cons); v1_ld = 4; v1_dst = 0; v1_cp = 1; }
    else { v1_base = reinterpret_cast<const float*>(P.nbr); v1_ld = 4; v1_dst = 4; v1_cp = 1; }      // lanes 29..31 (30, 31 do not park)
    const float* s0_base = P.eps_attr + ((size_t)b * A_ + bl) * G * G;                                  // eps_attr channel bl
    const int s0_dst = BD_EA + bl;
    if (bl < 18) { s1_base = P.eps_attr + ((size_t)b * A_ + 32 + bl) * G * G; s1_stride = 1; s1_row = 0; s1_dst = BD_EA + 32 + bl; }
    else if (bl < 22) { s1_base = P.eps_box + ((size_t)b * 4 + (bl - 18)) * G * G; s1_stride = 1; s1_row = 0; s1_dst = BD_EB + (bl - 18); }
    else if (bl == 22) { s1_base = P.eps_depth + (size_t)b * G * G; s1_stride = 1; s1_row = 0; s1_dst = BD_EPSD; }
    else if (bl == 23) { s1_base = P.rec + (REC - 1); s1_stride = L.ld_rec; s1_row = 1; s1_dst = BD_ZP; }
    else if (bl == 24) { s1_base = P.g_pres_r; s1_stride = 1; s1_row = 1; s1_dst = BD_GPR; }
    else if (bl == 25) { s1_base = P.Oo; s1_stride = L.ld_oo; s1_row = 1; s1_dst = BD_OO; }
    else if (bl == 26) { s1_base = P.g_depth_r; s1_stride = 1; s1_row = 1; s1_dst = BD_GDR; }
    else if (bl == 27) { s1_base = P.Oz + NP; s1_stride = L.ld_oz; s1_row = 1; s1_dst = BD_OZ0; }
    else { s1_base = P.Oz + NP + 1; s1_stride = L.ld_oz; s1_row = 1; s1_dst = BD_OZ1; }                 // lanes 28..31 (29..31 do not park)
    float4 pf_v0, pf_v1;
    float pf_s0, pf_s1;
    auto hlo_of = [&](int tt) { return max(0, (tt - G + 2) >> 1); };       // first grid row on diagonal tt (cells ordered by h)
    auto bundle_fetch = [&](int tn) {
        const int c0n = bc0_sh[tn], ncn = bnc_sh[tn];
        const int k = min(brow, ncn - 1);
        const int cpn = c0n + k, h = max(hlo_of(tn), hb0) + k, w = tn - 2 * h;
        const size_t rn = (size_t)cpn * L.B + b, cell = (size_t)h * G + w;
        pf_v0 = CH_GLOADF4(v0_base + rn * v0_ld);
        pf_v1 = CH_GLOADF4(v1_base + (v1_cp ? (size_t)cpn : rn) * v1_ld);
        pf_s0 = s0_base[cell];
        pf_s1 = s1_base[(s1_row ? rn : cell) * s1_stride];
    };
    auto bundle_park = [&]() {
        *reinterpret_cast<float4*>(&bundle_sh[brow][v0_dst]) = pf_v0;
        if (v1_cp) { if (bl < 30) *reinterpret_cast<float4*>(&ibundle_sh[brow][v1_dst]) = pf_v1; }
        else *reinterpret_cast<float4*>(&bundle_sh[brow][v1_dst]) = pf_v1;
        bundle_sh[brow][s0_dst] = pf_s0;
        if (bl < 29) bundle_sh[brow][s1_dst] = pf_s1;
    };
    bundle_fetch(t_last);
    bundle_park();
    __syncthreads();

    // ---- band split (see k_chain_fwd): the band BELOW publishes, for each cell of its first row, the context-gradient pieces of its UL / U /
    // UR slots -- the record gradients of three cells of this band's last row -- as soon as its BOX0 stage has completed the sums; this band
    // gathers the three pieces of its last-row cell one wavefront ahead.  Counter: wavefronts finished, counted from the last one.
    typedef unsigned long long u64_t;
    int sync_dead = 0;
    auto publish_grad = [&](int tq, const float (*sl)[LD_R], int ln) {       // store wave, behind the BOX0 barrier of wavefront tq
        if (NBn > 1 && band > 0) {
            const int w = tq - 2 * hb0;                                      // this band's first row on wavefront tq (its tile row 0), if present
            if (w >= 0 && w < G && hlo_of(tq) <= hb0) {
                for (int i = ln; i < 3 * REC / 2; i += 64) {
                    const int sidx = i / (REC / 2), jj = (i - sidx * (REC / 2)) * 2, wt = w - 1 + sidx;      // slot UL -> cell w-1, U -> w, UR -> w+1
                    if (wt >= 0 && wt < G) {
                        const u64_t v = *reinterpret_cast<const u64_t*>(&sl[0][F + sidx * REC + jj]);
                        __hip_atomic_store(reinterpret_cast<u64_t*>(a.bnd_grad + (((size_t)(b * NBn + band) * G + wt) * 3 + sidx) * REC + jj), v,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ln == 0) __hip_atomic_store(flag_mine, T - tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto fetch_grad = [&](int tq, int ln) {                                  // pieces for this band's last-row cell of wavefront tq -> bgrad_sh (whole wave)
        const int w = tq - 2 * (hb1 - 1);
        if (w < 0 || w >= G) return;                                         // (wave-uniform)
        const int need = T - max(tq + 1, 2 * hb1);                           // the band below has finished wavefront max(tq + 1, its first)
        const int* flag = a.sync + CHAIN_SYNC_HDR + L.B * NBn + b * NBn + (band + 1);
        int ok = sync_dead;
        for (int spin = 0; spin < CHAIN_SPIN_LIMIT && !ok; ++spin) {
            ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need);
            if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) { sync_dead = 1; if (ln == 0) { a.sync[2] = 1; a.sync[CHAIN_SYNC_STICKY(L.B, NBn)] = 1; } }
        for (int i = ln; i < 3 * REC / 2; i += 64) {
            const u64_t v = __hip_atomic_load(reinterpret_cast<const u64_t*>(a.bnd_grad + ((size_t)(b * NBn + band + 1) * G + w) * 3 * REC) + i,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<u64_t*>(&bgrad_sh[0][0] + 2 * i) = v;
        }
    };
    if (band < NBn - 1 && wave == 7) fetch_grad(t_last, lane);
    __syncthreads();

    WPipeB pipe;
    pipe_fill_w<4, 7>(a.wt[CW_OBJ1], pipe, wave, lane);      // the first data-gradient GEMM of the first wavefront
    int stamp_j = 2048;
#define CB_STAMP() do { if (a.stamps && b == 0 && band == 0 && tid0 == 0) a.stamps[stamp_j++] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int t = t_last; t >= t_first; --t) {
        CB_STAMP();
        tid = tid0;   // opaque per-iteration copy: no LICM of lane-dependent address arithmetic (see k_chain_fwd)
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        const int c0 = bc0_sh[t];
        const int nc = bnc_sh[t];
        float (*slot)[LD_R] = ring[t & 3];
        if (tid < MT) {
            const int k = min(tid, nc - 1);
            const int cp = c0 + k, h = max(hlo_of(t), hb0) + k;
            row_r[tid] = cp * L.B + b;
            row_h[tid] = h;
            row_w[tid] = t - 2 * h;
            zp_sh[tid] = bundle_sh[tid][BD_ZP];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cons_sh[tid][q] = ibundle_sh[tid][q];
                nbr_row[tid][q] = ibundle_sh[tid][4 + q];
                nb_sh[tid][q] = bundle_sh[tid][BD_NB + q];
            }
        }
        lds_barrier();
        CB_STAMP();
        // ---- prefetch everything this step needs from HBM that does not depend on the chain: relu masks of the 7 hidden layers and
        // the saved glimpse derivatives for this wave's tiles (consumed ~40 us later: their latency is fully hidden)
        // relu sign bits of this wavefront: one word per thread to park (threads 0..263), plus the four words that hold this thread's four
        // Ho2 elements for the rank-1 product of the presence stage
        const unsigned long long* const mbt = P.mbits + ((size_t)(b * NBn + band) * T + t) * (MB_TILES * 4);
        const unsigned long long mbq = mbt[min(tid, MB_TILES * 4 - 1)];
        const int ho_q4 = min(tid & 31, 24);          // the pres stage's mapping: 32 lanes per row, lanes 0..24 own 4 columns of Ho2 each
        // the four sign-bit words of this thread's columns 4*l .. 4*l+3 (tile l>>2, column group l&3: one word per column)
        typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
        const u64x2_t mbo01 = *reinterpret_cast<const u64x2_t*>(mbt + (MB_HO2 + (ho_q4 >> 2)) * 4);
        const u64x2_t mbo23 = *reinterpret_cast<const u64x2_t*>(mbt + (MB_HO2 + (ho_q4 >> 2)) * 4 + 2);
        // ---- B1a: gradient of each cell's record from its consumers' context columns (wavefronts t+1..t+3)
        // (branch-free: the four consumer slots are read side by side -- with a `continue` per slot the loop was twelve dependent LDS
        // round trips, 1.2 us per wavefront)
        {
            const int d1 = bc0_sh[min(t + 1, T)], d2 = bc0_sh[min(t + 2, T)], d3 = bc0_sh[min(t + 3, T)];
            const int hlast = band < NBn - 1 ? hb1 - 1 : -1;      // the grid row whose UL / U / UR consumers live in the band below
            const float (*r1)[LD_R] = ring[(t + 1) & 3];
            const float (*r2)[LD_R] = ring[(t + 2) & 3];
            const float (*r3)[LD_R] = ring[(t + 3) & 3];
            for (int idx = tid; idx < nc * REC; idx += NTH) {
                const int row = idx / REC, j = idx - row * REC;
                const int4 q = *reinterpret_cast<const int4*>(cons_sh[row]);
                const bool far = row_h[row] == hlast;
                const float v0 = far ? bgrad_sh[0][j] : r3[min(max(q.x - d3, 0), MT - 1)][F + j];
                const float v1 = far ? bgrad_sh[1][j] : r2[min(max(q.y - d2, 0), MT - 1)][F + REC + j];
                const float v2 = far ? bgrad_sh[2][j] : r1[min(max(q.z - d1, 0), MT - 1)][F + 2 * REC + j];
                const float v3 = r1[min(max(q.w - d1, 0), MT - 1)][F + 3 * REC + j];
                float g = q.x >= 0 ? v0 : 0.f;
                g += q.y >= 0 ? v1 : 0.f;
                g += q.z >= 0 ? v2 : 0.f;
                g += q.w >= 0 ? v3 : 0.f;
                grec[row][j] = g;
            }
        }
        lds_barrier();
        CB_STAMP();
        // saved glimpse derivatives for this wave's 7 ENC0 tiles in the (transposed) MFMA output layout itself -- row lane & 15, elements
        // tile*16 + (lane>>4)*4 .. +3: one 16-byte load per tile, and the epilogue needs no transpose through LDS any more.  Requested in three
        // instalments, in the stages that stream no weights -- three tiles here (behind the first stage's barrier), two at the depth stage, two at
        // the attribute stage -- not as one burst at the top of the wavefront, where seven more wave-loads held up the first stage's own issue
        // (timing-only removal of these loads: 0.733 -> 0.694 ms; all at the top 0.728, all here 0.719, 4 + 3 (depth) 0.690, 3 + 2 + 2: 0.686;
        // all at the OBJ0 stage, whose weight stream they then compete with, 0.757; all at the END of this stage 0.730)
        uint4 gxy_pf[7];
        auto gxy_fetch = [&](int j0, int j1) {
            const size_t grow = (size_t)row_r[min(lane & 15, nc - 1)] * L.ld_gl;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j >= j0 && j < j1) {
                    const int e0 = min((wave + NW * j) * 16 + (lane >> 4) * 4, GLN - 4);
                    gxy_pf[j] = CH_GLOAD16(P.gxy + grow + e0);
                }
        };
        gxy_fetch(0, 3);
        // ---- B1b: presence (32 threads per row: sum of the row's Gaussian KL elements, then d logit) and, by the same lanes, the rank-1
        // data gradient of the obj net's output layer: dHo2 = dOo (x) W_out masked by relu -- every lane of a row evaluates the row's d logit
        // (the reduction leaves the KL sum in all 32), lanes 0..24 produce 4 columns each from the sign-bit words they prefetched
        {
            const int row = tid >> 5, l = tid & 31;
            float kl = 0.f;
            if (row < nc) {
                const float* bd = bundle_sh[row];
                const float* st = bd + BD_ST;
                for (int j = l; j < A_; j += 32)
                    kl += ch_kl_gauss(bd[BD_OE + j], bd[BD_SD + j], prior_sh[4], prior_sh[12 + 4]);
                if (l < 4) kl += ch_kl_gauss(st[ST_MU_BOX + l], st[ST_SD_BOX + l], prior_sh[l], prior_sh[12 + l]);
                if (l == 4) kl += ch_kl_gauss(st[ST_MU_DEPTH], st[ST_SD_DEPTH], prior_sh[5], prior_sh[12 + 5]);
            }
            kl = dpp_add_<0xB1>(kl); kl = dpp_add_<0x4E>(kl); kl = dpp_add_<0x141>(kl); kl = dpp_add_<0x140>(kl);     // 16-lane rows (DPP)
            kl += __shfl_xor(kl, 16, 64);                                                                             // the two rows of a 32-lane group
            float d = 0.f;
            if (row < nc) {
                const float* bd = bundle_sh[row];
                // pres_backward (cell_math.h): d/dz of the Bernoulli KL term (models.py:223-226) + what arrives at z_pres, through the sigmoid
                const float z = zp_sh[row], pz = bd[BD_ST + ST_PZ], e = 1e-9f;
                const float dkl = ch_log(z + e) - ch_log(pz + e) + z * __builtin_amdgcn_rcpf(z + e) - ch_log(1.f - z + e) + ch_log(1.f - pz + e) -
                                  (1.f - z) * __builtin_amdgcn_rcpf(1.f - z + e);
                const float g = grec[row][REC - 1] + bd[BD_GPR] + ks * (kl + dkl);
                d = g * z * (1.f - z) * in10(freeze_val(H.wheel, bd[BD_OO])) * (1.f - H.wheel);
                if (l == 0) reinterpret_cast<__bf16*>(P.dOo)[(size_t)row_r[row] * L.ld_oo] = (__bf16)d;
            }
            if (l < 25) {
                const float4 w = *reinterpret_cast<const float4*>(&wobj_sh[l * 4]);
                const int hs = (l & 3) * 16 + row;                                                           // bit of (column group, row)
                bf16x4 o;
                o[0] = (__bf16)(((mbo01[0] >> hs) & 1ull) ? d * w.x : 0.f); o[1] = (__bf16)(((mbo01[1] >> hs) & 1ull) ? d * w.y : 0.f);
                o[2] = (__bf16)(((mbo23[0] >> hs) & 1ull) ? d * w.z : 0.f); o[3] = (__bf16)(((mbo23[1] >> hs) & 1ull) ? d * w.w : 0.f);
                if (row < nc) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dHo2) + (size_t)row_r[row] * SP_LDH + l * 4) = o;
                *reinterpret_cast<bf16x4*>(&Aa[row * LD_H + l * 4]) = o;
            }
        }
        if (tid < MB_TILES * 4) mb_sh[tid] = mbq;            // first consumer is behind the next barrier
        lds_barrier();
        CB_STAMP();
        // 7-tile layers: the 8th wave has no tile; the layer-output gradient (bf16 in LDS) is copied to HBM by that wave one stage later
        if (wave < 7) {
            wg_gemm_wt<4, 7>(Aa, LD_H, a.wt[CW_OBJ1], pipe, wave, lane,
                [&](int j, int nt, const f32x4& acc) { hidden_epi<7>(nt, acc, mb_sh + MB_HO1 * 4, SP_H, Ab, nullptr, 0, row_r, nc, lane); },
                [&]() { pipe_fill_w<4, 30>(a.wt[CW_OBJ0], pipe, wave, lane); });
        } else {
            pipe_fill_w<4, 30>(a.wt[CW_OBJ0], pipe, wave, lane);
        }
        lds_barrier();
        CB_STAMP();
        wg_gemm_wt<4, 30>(Ab, LD_H, a.wt[CW_OBJ0], pipe, wave, lane, [&](int j, int nt, const f32x4& acc) {
            const int n0 = nt * 16 + (lane >> 4) * 4, row = lane & 15;
            if (nt >= 30) return;                                    // wave-uniform: clamped tiles past the layer
            if (n0 < F + CTX) *reinterpret_cast<f32x4*>(&slot[row][n0]) = acc;
            else *reinterpret_cast<f32x4*>(&tailO[row][n0 - (F + CTX)]) = acc;
        }, [&]() { pipe_fill_w<4, 7>(a.wt[CW_ZH], pipe, wave, lane); });
        lds_barrier();
        CB_STAMP();
        gxy_fetch(3, 5);
        // ---- depth (models.py:88-97 backward); passthrough gradient -> z-net head
        if (wave == 7) copy_rows_w<25, 8>(Ab, LD_H * 2, P.dHo1, (size_t)SP_LDH * 2, row_r, nc, lane);      // OBJ1's output, untouched until the ZH stage
        for (int idx = tid; idx < MT * (NP / 4); idx += NTH) {       // 4 columns per thread: one pass over the 16 rows
            const int row = idx / (NP / 4), i = (idx - row * (NP / 4)) * 4;
            const float4 v = row < nc ? *reinterpret_cast<const float4*>(&tailO[row][i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const bf16x4 o = pack4(v.x, v.y, v.z, v.w);
            *reinterpret_cast<bf16x4*>(&Aa[row * LD_H + i]) = o;
            if (row < nc) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dOz) + (size_t)row_r[row] * L.ld_oz + i) = o;
        }
        if (tid < MT) {
            float d_mu = 0.f, d_ls = 0.f;
            if (tid < nc) {
                const size_t r = row_r[tid];
                const float* bd = bundle_sh[tid];
                const float* st = bd + BD_ST;
                const float eps = bd[BD_EPSD], mu = st[ST_MU_DEPTH], sd = st[ST_SD_DEPTH], ls = bd[BD_OZ1], zp = zp_sh[tid];
                const float g_depth = grec[tid][4 + A_] + tailO[tid][NP + 4 + A_] + bd[BD_GDR];
                // depth_backward (cell_math.h)
                const float dl = mu + sd * eps;
                const float sg = ch_sigmoid(clamp10(dl));
                const float g_dl = g_depth * 4.f * sg * (1.f - sg) * in10(dl);
                const float m = prior_sh[5], rs = prior_sh[12 + 5];
                d_mu = (g_dl + ks * zp * (mu - m) * rs * rs) * (1.f - H.wheel);
                const float g_sd = (g_dl * eps + ks * zp * (sd * rs * rs - __builtin_amdgcn_rcpf(sd))) * (1.f - H.wheel);
                const float sl = ch_sigmoid(clamp10(ls));
                d_ls = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
                reinterpret_cast<__bf16*>(P.dOz)[r * L.ld_oz + NP] = (__bf16)d_mu;
                reinterpret_cast<__bf16*>(P.dOz)[r * L.ld_oz + NP + 1] = (__bf16)d_ls;
            }
            Aa[tid * LD_H + NP] = (__bf16)d_mu;
            Aa[tid * LD_H + NP + 1] = (__bf16)d_ls;
        }
        lds_barrier();
        CB_STAMP();
        if (wave < 7) {
            wg_gemm_wt<4, 7>(Aa, LD_H, a.wt[CW_ZH], pipe, wave, lane,
                [&](int j, int nt, const f32x4& acc) { hidden_epi<7>(nt, acc, mb_sh + MB_HZ2 * 4, SP_H, Ab, nullptr, 0, row_r, nc, lane); },
                [&]() { pipe_fill_w<4, 7>(a.wt[CW_Z1], pipe, wave, lane); });
        } else if (band < NBn - 1 && t > t_first) {
            fetch_grad(t - 1, lane);          // (bgrad_sh's readers of this wavefront are behind the grec barrier)
        }
        lds_barrier();
        CB_STAMP();
        if (wave < 7) {
            wg_gemm_wt<4, 7>(Ab, LD_H, a.wt[CW_Z1], pipe, wave, lane,
                [&](int j, int nt, const f32x4& acc) { hidden_epi<7>(nt, acc, mb_sh + MB_HZ1 * 4, SP_H, Aa, nullptr, 0, row_r, nc, lane); },
                [&]() { pipe_fill_w<4, 30>(a.wt[CW_Z0], pipe, wave, lane); });
        } else {
            pipe_fill_w<4, 30>(a.wt[CW_Z0], pipe, wave, lane);
            copy_rows_w<25, 8>(Ab, LD_H * 2, P.dHz2, (size_t)SP_LDH * 2, row_r, nc, lane);
        }
        lds_barrier();
        CB_STAMP();
        wg_gemm_wt<4, 30>(Aa, LD_H, a.wt[CW_Z0], pipe, wave, lane, [&](int j, int nt, const f32x4& acc) {
            const int n0 = nt * 16 + (lane >> 4) * 4, row = lane & 15;
            if (nt >= 30) return;
            if (n0 < F + CTX) {
                f32x4* q = reinterpret_cast<f32x4*>(&slot[row][n0]);
                *q = *q + acc;
            } else {
                *reinterpret_cast<f32x4*>(&tailZ[row][n0 - (F + CTX)]) = acc;
            }
        }, [&]() { pipe_fill_w<4, 8>(a.wt[CW_ENC2], pipe, wave, lane); });
        lds_barrier();
        CB_STAMP();
        gxy_fetch(5, 7);
        // ---- attributes -> gradient of the encoder output
        if (wave == 7) copy_rows_w<25, 8>(Aa, LD_H * 2, P.dHz1, (size_t)SP_LDH * 2, row_r, nc, lane);      // Z1's output, untouched until the ENC2 stage
        for (int idx = tid; idx < MT * A_; idx += NTH) {
            const int row = idx / A_, j = idx - row * A_;
            float d_mean = 0.f, d_ls = 0.f;
            if (row < nc) {
                const size_t r = row_r[row];
                const float* bd = bundle_sh[row];
                const float g = grec[row][4 + j] + tailZ[row][NP + 4 + j] + tailO[row][NP + 4 + j] + bd[BD_GA + j];
                // attr_backward (cell_math.h)
                const float mu = bd[BD_OE + j], sd = bd[BD_SD + j], ls = bd[BD_OE + A_ + j], eps = bd[BD_EA + j], zp = zp_sh[row];
                const float m = prior_sh[4], rs = prior_sh[12 + 4];
                const float g_sd = g * eps + ks * zp * (sd * rs * rs - __builtin_amdgcn_rcpf(sd));
                const float sl = ch_sigmoid(clamp10(ls));
                d_mean = g + ks * zp * (mu - m) * rs * rs;
                d_ls = g_sd * 2.f * sl * (1.f - sl) * in10(ls);
                reinterpret_cast<__bf16*>(P.dOe)[r * L.ld_oe + j] = (__bf16)d_mean;
                reinterpret_cast<__bf16*>(P.dOe)[r * L.ld_oe + A_ + j] = (__bf16)d_ls;
            }
            Ab[row * LD_H + j] = (__bf16)d_mean;
            Ab[row * LD_H + A_ + j] = (__bf16)d_ls;
        }
        lds_barrier();
        CB_STAMP();
        wg_gemm_wt<4, 8>(Ab, LD_H, a.wt[CW_ENC2], pipe, wave, lane,
            [&](int j, int nt, const f32x4& acc) { hidden_epi<8>(nt, acc, mb_sh + MB_HE2 * 4, SP_ENC_H2, Aa, reinterpret_cast<__bf16*>(P.dHe2), SP_ENC_H2, row_r, nc, lane); },
            [&]() { pipe_fill_w<4, 16>(a.wt[CW_ENC1], pipe, wave, lane); });
        lds_barrier();
        CB_STAMP();
        wg_gemm_wt<4, 16>(Aa, LD_H, a.wt[CW_ENC1], pipe, wave, lane,
            [&](int j, int nt, const f32x4& acc) { hidden_epi<16>(nt, acc, mb_sh + MB_HE1 * 4, SP_ENC_H1, Ab, reinterpret_cast<__bf16*>(P.dHe1), SP_ENC_H1, row_r, nc, lane); },
            [&]() { pipe_fill_w<8, 49>(a.wt[CW_ENC0], pipe, wave, lane); });
        lds_barrier();
        CB_STAMP();
        // ---- d glimpse -> d z_where inside the epilogue (stn backward, modules.py:216-273): the glimpse gradient is never stored;
        // each element meets the (d val/d gx, d val/d gy) pair the forward kernel saved -- prefetched in this very layout -- lane sums are
        // reduced once per layer
        {
            float gs[4] = {0.f, 0.f, 0.f, 0.f};
            const bool live = (lane & 15) < nc;
            wg_gemm_wt<8, 49>(Ab, LD_H, a.wt[CW_ENC0], pipe, wave, lane, [&](int j, int nt, const f32x4& acc) {
                if (nt >= 49 || j >= 7) return;                           // wave-uniform: tiles past the glimpse
                const int e0 = nt * 16 + (lane >> 4) * 4;
                const int gi = e0 / PG, gj0 = e0 - gi * PG;             // PG % 4 == 0: the 4 elements share the glimpse row gi
                const float Y = pbase_sh[gi];
                const unsigned int pkq[4] = {gxy_pf[j].x, gxy_pf[j].y, gxy_pf[j].z, gxy_pf[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    union { unsigned int u; _Float16 h[2]; } pk;
                    pk.u = pkq[q];
                    const float d = live ? acc[q] : 0.f;
                    const float gix = d * (float)pk.h[0], giy = d * (float)pk.h[1];
                    gs[0] += gix; gs[1] += giy; gs[2] = fmaf(gix, pbase_sh[gj0 + q], gs[2]); gs[3] = fmaf(giy, Y, gs[3]);
                }
            }, [&]() { pipe_fill_w<4, 7>(a.wt[CW_BOXH], pipe, wave, lane); });
            // the four column groups of a row: lanes l, l + 16, l + 32, l + 48 (fixed association)
#pragma unroll
            for (int k = 0; k < 4; ++k) { gs[k] += __shfl_xor(gs[k], 16, 64); gs[k] += __shfl_xor(gs[k], 32, 64); }
            if (lane < 16 && live)                                                                                   // tx = 2*xt - 1
                *reinterpret_cast<float4*>(&gnbw[wave][lane][0]) = make_float4(2.f * gs[0], 2.f * gs[1], gs[2], gs[3]);
        }
        lds_barrier();
        CB_STAMP();
        // next wavefront's bundle, parked in the last stage: requested here, in a stage that issues no other loads, with ~3 us to land --
        // not in the top-of-wavefront burst, where its four loads held up the first stage (kernel 0.714 -> 0.698 ms; at the depth stage 0.704,
        // at the attribute stage 0.709, with the glimpse derivatives behind the first barrier 0.721)
        bundle_fetch(max(t - 1, t_first));
        // ---- box (models.py:322-381 backward); passthrough gradient -> box-net head
        for (int idx = tid; idx < MT * (NP / 4); idx += NTH) {       // 4 columns per thread: one pass over the 16 rows
            const int row = idx / (NP / 4), i = (idx - row * (NP / 4)) * 4;
            const float4 v = row < nc ? *reinterpret_cast<const float4*>(&tailZ[row][i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const bf16x4 o = pack4(v.x, v.y, v.z, v.w);
            *reinterpret_cast<bf16x4*>(&Aa[row * LD_H + i]) = o;
            if (row < nc) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dOb) + (size_t)row_r[row] * L.ld_ob + i) = o;
        }
        // one latent per lane (threads 64 .. 64 + 4*MT, wave 1: wave 0 runs the copy loop above): lane k of a row owns z_k, i.e. the
        // gradients of box / nbox element k ^ 1 and the latent's mean and log-std (box_backward, cell_math.h, element by element)
        if (tid >= 64 && tid < 64 + 4 * MT) {
            const int row = (tid - 64) >> 2, k = (tid - 64) & 3, o = k ^ 1;
            float d_mu = 0.f, d_ls = 0.f;
            if (row < nc) {
                const size_t r = row_r[row];
                const float* bd = bundle_sh[row];
                const float* st = bd + BD_ST;
                float gn = gnbw[0][row][o];
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) gn += gnbw[wv][row][o];
                gn += bd[BD_GNB + o];
                const float gb = grec[row][o] + tailZ[row][NP + o] + tailO[row][NP + o];
                const float mu = st[ST_MU_BOX + k], sd = st[ST_SD_BOX + k], eps = bd[BD_EB + k], lls = bd[BD_OBL + k], zp = zp_sh[row];
                const float gq = k < 2 ? (gb + gn * H.cell_over_img) * H.range_yx : (gb + gn * H.anchor / H.img) * H.range_hw;
                const float z = mu + sd * eps;
                const float sg = ch_sigmoid(clamp10(z));
                const float g_z = gq * sg * (1.f - sg) * in10(z);
                const float m = prior_sh[k], rs = prior_sh[12 + k];
                d_mu = (g_z + ks * zp * (mu - m) * rs * rs) * (1.f - H.wheel);
                const float g_sd = (g_z * eps + ks * zp * (sd * rs * rs - __builtin_amdgcn_rcpf(sd))) * (1.f - H.wheel);
                const float sl = ch_sigmoid(clamp10(lls));
                d_ls = g_sd * 2.f * sl * (1.f - sl) * in10(lls);
                reinterpret_cast<__bf16*>(P.dOb)[r * L.ld_ob + NP + k] = (__bf16)d_mu;
                reinterpret_cast<__bf16*>(P.dOb)[r * L.ld_ob + NP + 4 + k] = (__bf16)d_ls;
            }
            Aa[row * LD_H + NP + k] = (__bf16)d_mu;
            Aa[row * LD_H + NP + 4 + k] = (__bf16)d_ls;
        }
        lds_barrier();
        CB_STAMP();
        if (wave < 7) {
            wg_gemm_wt<4, 7>(Aa, LD_H, a.wt[CW_BOXH], pipe, wave, lane,
                [&](int j, int nt, const f32x4& acc) { hidden_epi<7>(nt, acc, mb_sh + MB_HB2 * 4, SP_H, Ab, nullptr, 0, row_r, nc, lane); },
                [&]() { pipe_fill_w<4, 7>(a.wt[CW_BOX1], pipe, wave, lane); });
        }
        lds_barrier();
        CB_STAMP();
        if (wave < 7) {
            wg_gemm_wt<4, 7>(Ab, LD_H, a.wt[CW_BOX1], pipe, wave, lane,
                [&](int j, int nt, const f32x4& acc) { hidden_epi<7>(nt, acc, mb_sh + MB_HB1 * 4, SP_H, Aa, nullptr, 0, row_r, nc, lane); },
                [&]() { pipe_fill_w<4, 21>(a.wt[CW_BOX0], pipe, wave, lane); });
        } else {
            pipe_fill_w<4, 21>(a.wt[CW_BOX0], pipe, wave, lane);
            copy_rows_w<25, 8>(Ab, LD_H * 2, P.dHb2, (size_t)SP_LDH * 2, row_r, nc, lane);
        }
        lds_barrier();
        CB_STAMP();
        wg_gemm_wt<4, 21>(Aa, LD_H, a.wt[CW_BOX0], pipe, wave, lane, [&](int j, int nt, const f32x4& acc) {
            const int n0 = nt * 16 + (lane >> 4) * 4, row = lane & 15;
            if (nt < 21 && n0 < F + CTX) {
                f32x4* q = reinterpret_cast<f32x4*>(&slot[row][n0]);
                *q = *q + acc;
            }
        }, [&]() { pipe_fill_w<4, 7>(a.wt[CW_OBJ1], pipe, wave, lane); });      // the next wavefront's first GEMM
        lds_barrier();
        CB_STAMP();
        // ---- d feat out; out-of-grid context slots feed the learned edge element
        if (wave == 7) {
            copy_rows_w<25, 8>(Aa, LD_H * 2, P.dHb1, (size_t)SP_LDH * 2, row_r, nc, lane);      // BOX1's output
            publish_grad(t, slot, lane);
        }
        // four columns per thread (F and REC are multiples of 4: a quad never straddles the feature / neighbour-slot boundaries)
        for (int idx = tid; idx < nc * (F / 4); idx += NTH) {
            const int row = idx / (F / 4), n = (idx - row * (F / 4)) * 4;
            const float4 v = *reinterpret_cast<const float4*>(&slot[row][n]);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(P.dfeat16) + ((size_t)(b * G + row_h[row]) * G + row_w[row]) * P.ld_feat + n) =
                pack4(v.x, v.y, v.z, v.w);                                                          // read by the 1x1 stack
        }
        // out-of-grid context slots feed the edge element: thread (s, j) owns element j of neighbour slot s and adds the wavefront's rows
        // in row order into a REGISTER that lives across the wavefronts (LDS atomics from the quad loop above gave sums that differed in
        // the last bits from run to run; a single owner thread walking all (row, slot) pairs cost ~1 us per wavefront)
        if (tid >= NTH / 2 && tid < NTH / 2 + 4 * REC) {
            const int es = (tid - NTH / 2) / REC, ej = (tid - NTH / 2) - es * REC;
            float add = 0.f;
            for (int row = 0; row < nc; ++row) {
                const float v = slot[row][F + es * REC + ej];
                add += nbr_row[row][es] < 0 ? v : 0.f;
            }
            edge_reg += add;
        }
        bundle_park();                       // every reader of this wavefront's bundle is behind the BOX0 barrier
        // every register a prefetch load of this wavefront wrote is consumed here on EVERY path: a load the compiler cannot prove complete at
        // the loop's back edge makes it guard the next definition of that register with s_waitcnt vmcnt(0) -- at the top of the next
        // wavefront that was a wait for the acknowledgement of every store of this stage (the 2.3 us "grec" stage of round 3's stamps)
#pragma unroll
        for (int j = 0; j < 7; ++j) asm volatile("" :: "v"(gxy_pf[j].x), "v"(gxy_pf[j].y), "v"(gxy_pf[j].z), "v"(gxy_pf[j].w));
        asm volatile("" :: "v"(mbq), "v"(mbo01), "v"(mbo23));
        lds_barrier();
        CB_STAMP();
    }
    // per-(sample, neighbour slot) partials of the edge element's gradient; chain_edge_reduce adds them in a fixed order
    if (tid >= NTH / 2 && tid < NTH / 2 + 4 * REC) a.gedge_part[(size_t)(b * NBn + band) * 4 * REC + (tid - NTH / 2)] = edge_reg;
}

// gedge[j] += sum_b sum_s part[b][s][j]: 16 row groups of one workgroup sum contiguous chunks in row order (8 loads in flight each), then the
// 16 group sums are added in group order -- the same association on every run
__global__ __launch_bounds__(1024) void k_edge_reduce(const float* __restrict__ part, int B, float* __restrict__ gedge,
                                                      const int* __restrict__ failed) {
    __shared__ float grp[16][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int n = 4 * B, per = (n + 15) / 16;
    const int lo = g * per, hi = min(n, lo + per);
    float t = 0.f;
    if (j < REC) {
        int i = lo;
        for (; i + 8 <= hi; i += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = part[(size_t)(i + e) * REC + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) t += v[e];
        }
        for (; i < hi; ++i) t += part[(size_t)i * REC + j];
    }
    grp[g][j] = t;
    __syncthreads();
    if (g == 0 && j < REC) {
        float r = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) r += grp[e][j];
        // a band-split hand-off of this workspace timed out (sticky word): the gradients of this step are not to be used -- NaN into a
        // parameter gradient makes that visible in the next loss without a host-side check
        gedge[j] += (failed && *failed) ? __builtin_nanf("") : r;
    }
}

// G <= 32 (chain_fwd_supported): ceil(G / 8) bands of as-even-as-possible height, at most CHAIN_MAX_BANDS
int chain_bands(const SpairDims& d) { return d.G > 16 ? (d.G + 7) / 8 : 1; }
static_assert((32 + 7) / 8 <= CHAIN_MAX_BANDS, "chain_fwd_supported() admits G <= 32");

static int chain_sync_reset(const ChainArgs& a, hipStream_t s) {
    if (a.nbands <= 1) return SPAIR_OK;
    if (!a.sync || !a.bnd_rec || !a.bnd_grad) return SPAIR_ERR_SHAPE;
    const size_t n = (size_t)CHAIN_SYNC_HDR + 2 * (size_t)a.L.B * a.nbands;
    return hipMemsetAsync(a.sync, 0, n * sizeof(int), s) == hipSuccess ? SPAIR_OK : SPAIR_ERR_LAUNCH;
}

int chain_bwd(const ChainArgs& a, hipStream_t s) {
    if (!a.gedge_part) return SPAIR_ERR_SHAPE;
    if (chain_sync_reset(a, s) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_chain_bwd, dim3(a.L.B * a.nbands), dim3(NTH), 0, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
int chain_edge_reduce(const ChainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_reduce, dim3(1), dim3(1024), 0, s, a.gedge_part, a.L.B * a.nbands, a.gedge,
                       a.nbands > 1 && a.sync ? a.sync + CHAIN_SYNC_STICKY(a.L.B, a.nbands) : nullptr);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}

int chain_fwd_supported(const SpairDims& d) {
    return d.dtype == SPAIR_BF16 && d.lookback <= 1 && !d.obj_conv && d.F == F && d.A == A_ && d.NP == NP && d.P == 28 && d.C == 1 && d.G <= 32 &&
           (d.G + 1) / 2 <= MT && d.G >= 2;
}

int chain_image_fp16(const SpairDims& d) { return chain_fwd_supported(d) && d.I <= IMG_MAX && (d.I * d.I) % 4 == 0; }

int chain_fwd(const ChainArgs& a, hipStream_t s) {
    if (chain_sync_reset(a, s) != SPAIR_OK) return SPAIR_ERR_LAUNCH;
    if (a.I <= IMG_MAX && (a.I * a.I) % 4 == 0) hipLaunchKernelGGL(k_chain_fwd<true>, dim3(a.L.B * a.nbands), dim3(NTH), 0, s, a);
    else hipLaunchKernelGGL(k_chain_fwd<false>, dim3(a.L.B * a.nbands), dim3(NTH), 0, s, a);
    SPAIR_CHECK_LAUNCH();
    return SPAIR_OK;
}
